// The scan kernel: geometry, decode, filters, seed-and-verify, sk_scan_grid -- part of sk_device.hip (included there, in this order; not a translation unit of its own).

// ---------------------------------------------------------------------------------------------
// scan kernel geometry
// ---------------------------------------------------------------------------------------------
#ifndef SK_THREADS
#define SK_THREADS      256                 // 4 waves of 64
#endif
#define SK_WAVES        (SK_THREADS / 64)
static_assert(SK_THREADS <= 256 && SK_THREADS % 64 == 0, "a tile's window numbers are kept in 16 bits (wq, cq): at most 256 threads x 128 positions");
#define SK_SPAN         128                 // window-end positions per thread
#define SK_SPAN_CH      8                   // 16-base chunks per span
#define SK_TILE         (SK_THREADS * SK_SPAN)
#define SK_NREC         (SK_THREADS + 1)    // record 0 = the 128 bases in front of the tile
#define SK_REC_DW       12                  // per record: 8 code words (u32) + 8 invalid masks (u16) = 48 B;
                                            // 12-dword lane stride keeps ds_read_b128 conflict-free
#define SK_NCHUNK       (SK_NREC * SK_SPAN_CH)
#ifndef SK_STREAM_POLICY
#define SK_STREAM_POLICY 0
#endif
#define SK_NCHUNK_GRID  (SK_NCHUNK + 1)
#ifndef SK_AGG_LOG2
#define SK_AGG_LOG2     8
#endif
#define SK_AGG          (1u << SK_AGG_LOG2) // per-workgroup table of rows already counted in the tile
#define SK_UNION_EAGER  16384u              // union tally: records and log entries that travel back with the counters (320 KiB of page-locked memory)
#define SK_EV_PAIRS     64u                 // launches whose timing events are kept before they are added up
#define SK_ODDCAP       (1u << 20)          // list of chunks with odd bytes; beyond it the byte-string kernel scans everything     // grid kernel: plus the chunk after the tile
#ifndef SK_CHUNK_REJECT
#define SK_CHUNK_REJECT 1                   // stage 2: three filter questions per differing base before its ~31 windows go one by one
#endif
#ifndef SK_ANCHOR_CH
#define SK_ANCHOR_CH    8u                  // stage 2: one table probe per this many consecutive surviving chunks (and the first); 2: -10 %, 4: -1 %
#endif
//                 // stage 2: one hash probe per this many consecutive windows

typedef uint32_t sk_u4 __attribute__((ext_vector_type(4)));

struct sk_table_view {
    const sk_u4    *slots;           // {key lo, key hi, counter index, text position << 1 | 1 if the key is strain text at its first occurrence}
    uint32_t        mask;
    uint32_t        nrows;
    // seed and verify: the strain's text (2 bits per base, 16 per word, first base on top) and, per 64 text
    // positions, {counter index of the first row that starts in the block, bit p: a row's first all-ACGT
    // occurrence starts at block + p, -}.  NULL: every window is probed on its own.
    const uint32_t *text2;
    const sk_u4    *rank;
    uint32_t        text_bases;
    // grid kernel: Bloom sets of the canonical 16-mers of the strain, a small one for the L2 and a
    // large one that settles what the small one lets through
    const uint2    *grid1, *grid2;
    uint32_t        grid1_blocks, grid2_shift;
    // chunks (stream offset / 16) in which phase 1 saw a byte that only the byte-string kernel can judge;
    // the count is flags[2]
    uint32_t       *oddlist;
    uint32_t        oddcap;
};

__device__ __forceinline__ uint64_t sk_slot_key(const sk_u4 e) { return ((uint64_t)e.y << 32) | e.x; }

// What a hit does.  COUNT mode (kmer_scrub_count): bump the row's counter in the scanned column.
// TALLY mode (strain_detect, src/strain_detect.c:477-485): bump the per-RECORD tallies (all hits /
// hits on rows whose type column holds `inf_value`) and log the latter as (position, row).
struct sk_sink {
    uint32_t       *counts;        // COUNT: counts + col * nrows
    uint32_t       *diff;          // COUNT: difference array of the scanned column [nrows + 1] (sk_diff_flush folds it in)
    const uint32_t *rec_start;     // TALLY: batch offset of every record's first byte, ascending
    const uint32_t *tile_first;    // TALLY: per 32768-byte tile, index of the first record starting in or after it
    uint32_t        nrec;
    uint32_t       *tally;         // TALLY: [2 * nrec]
    const uint32_t *type;          // TALLY: type column
    const uint32_t *infbits;       // TALLY: bit i <=> type[i] == inf_value (by counter index; two spare words behind)
    uint32_t        inf_value;
    const uint32_t *inv;           // TALLY: counter index -> caller's row (NULL = identity)
    uint2          *hits;          // TALLY: (window-end offset in batch, caller's row)
    unsigned long long *nhits;
    unsigned long long  hits_cap;
    // TALLY: the workgroup's share of the hit log is gathered in LDS (lds_hits[SK_AGG], filled up to *lds_n) and goes out
    // with ONE atomic on nhits at the end of the tile: a returning atomic on a single word per wave and batch was the
    // tail of every launch (same-address atomics serialise in the L2)
    uint2              *lds_hits;
    uint32_t           *lds_n;        // [0] places reserved so far, [1] end of the valid share (SK_AGG until a reservation did not fit)
    // TALLY against a union table (sk_union: the key sets of `ns` strains in one table).  umask[row] = {bit s: strain s holds
    // the row's key, bit s: and it is informative there}; the tallies are then per (record, strain): tally[2 * (record * ns + s)],
    // and a log entry is (position | s << 26, row) -- one per strain in which the hit is informative.  ns != 0 says so, and the
    // masks then stand where the type column would (`type`, see sk_umask): the kernel is short of scalar registers.
    uint32_t            ns;
};
__device__ __forceinline__ const uint2 *sk_umask(const sk_sink &k) { return (const uint2 *)k.type; }
// union table: one byte per record, set when any of its (record, strain) tallies was touched -- the compaction behind the scan
// then reads (and zeroes again) only the rows of records that were hit, instead of the whole records x strains array.  It stands
// where the bitmap of informative rows would (`infbits`; the union has the masks for that): no scalar register to spare.
__device__ __forceinline__ uint8_t *sk_uflag(const sk_sink &k) { return (uint8_t *)const_cast<uint32_t *>(k.infbits); }

// union table: a hit of `count` windows on a key held by the strains in `members`, in record `rec`
__device__ __forceinline__ void sk_union_credit(const sk_sink &k, uint32_t rec, uint32_t members, uint32_t count)
{
    if (members) sk_uflag(k)[rec] = 1;
    while (members) {
        const uint32_t s = (uint32_t)__builtin_ctz(members);
        members &= members - 1u;
        atomicAdd(&k.tally[2u * (rec * k.ns + s)], count);
    }
}
// ... and its informative side: one tally per strain in `infm` (the LOG gets one entry per hit, (position, global row), whatever
// the number of strains: sk_union_post deals it out to the strains behind the scan)
__device__ __forceinline__ void sk_union_informative(const sk_sink &k, uint32_t rec, uint32_t infm)
{
    while (infm) {
        const uint32_t s = (uint32_t)__builtin_ctz(infm);
        infm &= infm - 1u;
        atomicAdd(&k.tally[2u * (rec * k.ns + s) + 1u], 1u);
    }
}

// reserve `n` consecutive places of the hit log for the calling wave (wave-uniform n > 0; every lane gets the answer):
// in the workgroup's LDS share if they fit (*in_lds = true, index into lds_hits), else in the global log
__device__ __forceinline__ unsigned long long sk_log_reserve(const sk_sink &k, uint32_t n, uint32_t lane, bool *in_lds)
{
    uint32_t lb = 0;
    if (lane == 0u) lb = atomicAdd(k.lds_n, n);
    lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)lb);
    if (lb + n <= SK_AGG) { *in_lds = true; return lb; }
    // did not fit, and nothing reserved after it will (the counter only grows): the LDS share ends where this
    // reservation began -- remember the lowest such place -- and this wave's entries go to the global log
    if (lane == 0u) atomicMin(k.lds_n + 1, lb);
    unsigned long long base = 0;
    if (lane == 0u) base = atomicAdd(k.nhits, (unsigned long long)n);
    *in_lds = false;
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
}
__device__ __forceinline__ void sk_log_put(const sk_sink &k, bool in_lds, unsigned long long at, uint2 e)
{
    if (in_lds) k.lds_hits[at] = e;
    else if (at < k.hits_cap) k.hits[at] = e;
}

// TALLY: the record that holds batch offset pos = the last one whose start is <= pos.  The per-tile index narrows it to the
// records that start in pos's 32 KiB tile (or run into it); among those the starts are close to evenly spaced (reads of
// similar length), so an interpolated guess checked against its two neighbours usually settles it in two round trips
// instead of the eight of a binary search -- which finishes the job when the guess is off.
__device__ __forceinline__ uint32_t sk_record_of(const sk_sink &k, uint32_t pos)
{
    const uint32_t t = pos >> 15;
    uint32_t lo = k.tile_first[t], hi = k.tile_first[t + 1u];
    lo = lo ? lo - 1u : 0u;
    if (hi <= lo) hi = lo + 1u;
    if (hi - lo > 3u) {
        const uint32_t s_lo = k.rec_start[lo], s_hi = k.rec_start[hi - 1u];
        if (pos >= s_hi) return hi - 1u;
        uint32_t g = lo + (uint32_t)((float)(pos - s_lo) * (float)(hi - 1u - lo) / (float)(s_hi - s_lo));
        g = g > hi - 2u ? hi - 2u : g;
        const uint32_t a = k.rec_start[g], b = k.rec_start[g + 1u];
        if (a <= pos) { if (pos < b) return g; lo = g + 1u; } else hi = g;            // (lo stays a record with start <= pos)
    }
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (k.rec_start[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool TALLY, bool NOATOMIC = false, bool UNION = false>
__device__ __forceinline__ void sk_on_hit(const sk_sink &k, uint32_t row, uint32_t pos)
{
    if (NOATOMIC) { if (row == 0x7FFFFFFFu) k.counts[0] = pos; return; }      // timing experiment only
    if (!TALLY) { atomicAdd(&k.counts[row], 1u); return; }
    const uint32_t lo = sk_record_of(k, pos);
    if (UNION) {                                                    // (lane by lane: only sk_scan_wide comes this way)
        const uint2 mk = sk_umask(k)[row];
        sk_union_credit(k, lo, mk.x, 1u);
        if (mk.y) {
            sk_union_informative(k, lo, mk.y);
            const unsigned long long i = atomicAdd(k.nhits, 1ull);
            if (i < k.hits_cap) k.hits[i] = make_uint2(pos, row);
        }
        return;
    }
    atomicAdd(&k.tally[2u * lo], 1u);
    if (k.type[row] == k.inf_value) {
        atomicAdd(&k.tally[2u * lo + 1u], 1u);
        const unsigned long long i = atomicAdd(k.nhits, 1ull);
        if (i < k.hits_cap) k.hits[i] = make_uint2(pos, k.inv ? k.inv[row] : row);
    }
}

// TALLY for a whole wave at once (every lane calls it; `hit` = counter index or 0xFFFFFFFF): lanes of
// one read sit next to each other, so each run of lanes with the same record adds its hit counts
// with ONE atomic per tally word, and the hit log takes one atomic per wave.
template <bool UNION = false>
__device__ __forceinline__ void sk_tally_wave(const sk_sink &k, uint32_t hit, uint32_t pos, uint32_t lane)
{
    const bool is_hit = hit != 0xFFFFFFFFu;
    if (UNION) {
        // union table: per (record, strain).  Lanes of one read are neighbours and mostly name the same strains: a run of lanes
        // with the same record and the same members adds its hits with one atomic per member; the log takes (position, global
        // row) once per informative hit, gathered in LDS like the single strain's
        uint32_t rec = 0xFFFFFF00u | lane, mx = 0u, my = 0u;
        if (is_hit) {
            rec = sk_record_of(k, pos);
            const uint2 mk = sk_umask(k)[hit];
            mx = mk.x; my = mk.y;
        }
        const uint32_t prev_r = (uint32_t)__shfl_up((int)rec, 1), prev_m = (uint32_t)__shfl_up((int)mx, 1);
        const bool first = (lane == 0u) | (rec != prev_r) | (mx != prev_m);
        const unsigned long long fm = __ballot(first), hm = __ballot(is_hit), im = __ballot(my != 0u);
        if (first & is_hit) {
            const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
            const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
            const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
            sk_union_credit(k, rec, mx, (uint32_t)__popcll(hm & seg));
        }
        if (my) sk_union_informative(k, rec, my);
        if (im) {                                                  // (wave-uniform)
            bool in_lds;
            const unsigned long long base = sk_log_reserve(k, (uint32_t)__popcll(im), lane, &in_lds);
            if (my) sk_log_put(k, in_lds, base + (unsigned long long)__popcll(im & ((1ull << lane) - 1ull)), make_uint2(pos, hit));
        }
        return;
    }
    uint32_t rec = 0xFFFFFF00u | lane;                    // distinct per lane when there is no hit
    bool is_inf = false;
    if (is_hit) {
        rec = sk_record_of(k, pos);
        is_inf = k.type[hit] == k.inf_value;
    }
    const uint32_t prev = (uint32_t)__shfl_up((int)rec, 1);
    const bool first = (lane == 0u) | (rec != prev);
    const unsigned long long fm = __ballot(first), hm = __ballot(is_hit), im = __ballot(is_inf);
    if (first & is_hit) {
        const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
        const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
        const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
        atomicAdd(&k.tally[2u * rec], (uint32_t)__popcll(hm & seg));
        const uint32_t ni = (uint32_t)__popcll(im & seg);
        if (ni) atomicAdd(&k.tally[2u * rec + 1u], ni);
    }
    if (im) {                                                      // (wave-uniform)
        bool in_lds;
        const unsigned long long base = sk_log_reserve(k, (uint32_t)__popcll(im), lane, &in_lds);
        if (is_inf) sk_log_put(k, in_lds, base + (unsigned long long)__popcll(im & ((1ull << lane) - 1ull)), make_uint2(pos, k.inv ? k.inv[hit] : hit));
    }
}

// exact lookup: counter index of `canon`, or 0xFFFFFFFF; *is_text = the slot's orientation bit
__device__ __forceinline__ uint32_t sk_find(uint64_t canon, const sk_table_view &t, uint32_t *is_text)
{
    uint32_t slot = sk_slot0(sk_khash(canon), t.mask);
    for (;;) {
        const sk_u4 e = t.slots[slot];
        const uint64_t key = sk_slot_key(e);
        if (key == canon) { *is_text = e.w & 1u; return e.z; }
        if (key == SK_EMPTY64) return 0xFFFFFFFFu;
        slot = (slot + 1u) & t.mask;
    }
}

// stage 2 for one window: slot from the k-mer hash, linear probing, 62-bit compare
template <bool TALLY, bool NOATOMIC = false, bool UNION = false>
__device__ __forceinline__ void sk_probe(uint64_t canon, const sk_table_view &t, const sk_sink &k, uint32_t pos)
{
    uint32_t slot = sk_slot0(sk_khash(canon), t.mask);
    for (;;) {
        const sk_u4 e = t.slots[slot];
        const uint64_t key = sk_slot_key(e);
        if (key == canon) { sk_on_hit<TALLY, NOATOMIC, UNION>(k, e.z, pos); return; }
        if (key == SK_EMPTY64) return;
        slot = (slot + 1u) & t.mask;
    }
}

// ---- phase 1 helpers: 4 bytes at a time (SWAR) -------------------------------------------------
// bit 7 of every byte of the result is set iff that byte of x is non-zero
__device__ __forceinline__ uint32_t sk_nz_msb(uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; }

// w = 4 stream bytes (first base in the low byte).  codes8: their 2-bit codes, first base in bits
// 7..6.  inv4: bit i set iff byte i is not A/C/G/T (any case).
__device__ __forceinline__ void sk_decode4(uint32_t w, uint32_t &codes8, uint32_t &inv4)
{
    // The low three bits of a letter tell A (1), C (3), T (4), G (7) apart in either case: two byte permutes with them as
    // selector give the 2-bit code and the upper-case letter the byte would have to be; the byte is valid iff it IS that
    // letter (every other selector value expects 0xFF, which no upper-cased byte equals).  Codes of invalid bytes are 0:
    // no window that is looked at holds one.
    const uint32_t u   = w & 0xDFDFDFDFu;                                  // upper-cased letters
    const uint32_t sel = w & 0x07070707u;
    const uint32_t cd  = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, sel);   // A0 C1 G2 T3
    codes8 = (cd * 0x40100401u) >> 24;                                     // gather 4 x 2 bits
    const uint32_t d   = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, sel) ^ u;   // 0 <=> the byte is that letter
    inv4 = ((sk_nz_msb(d) & 0x80808080u) * 0x00204081u) >> 28;             // bit i <=> byte i is non-zero (the four top bits gathered)
}

// The same, and `odd` |= a word that is non-zero iff one of the four bytes is neither A/C/G/T (any case) nor N/n nor '\n' -- a byte only
// the exact byte-string kernel can judge.  One more permute with the same selector: '\n' (low bits 2) and N (6) get their own
// expected byte.  For that the case fold must spare what is no letter ('*' is 0x2A = '\n' + bit 5): bit 5 is cleared only where bit 6
// is set -- which changes nothing for A/C/G/T, whose expected bytes all have bit 6.  (Round 3: the kernel is bound by vector
// instructions since the wave-priority change, and the loop this replaces -- sk_chunk_has_odd_byte, 21 instructions per invalid byte of
// the busiest lane, entered for nearly every chunk because some lane of the wave always holds a '\n' -- was a seventh of them.)
__device__ __forceinline__ void sk_decode4o(uint32_t w, uint32_t &codes8, uint32_t &inv4, uint32_t &odd)
{
    const uint32_t u   = w & ~((w >> 1) & 0x20202020u);                    // letters upper-cased, everything else as it is
    const uint32_t sel = w & 0x07070707u;
    const uint32_t cd  = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, sel);   // A0 C1 G2 T3
    codes8 = (cd * 0x40100401u) >> 24;                                     // gather 4 x 2 bits
    const uint32_t d   = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, sel) ^ u;   // 0 <=> the byte is that letter
    inv4 = ((sk_nz_msb(d) & 0x80808080u) * 0x00204081u) >> 28;             // bit i <=> byte i is non-zero (the four top bits gathered)
    odd |= __builtin_amdgcn_perm(0x474EFF54u, 0x430A41FFu, sel) ^ u;       // 0 <=> every byte is one of A C G T a c g t N n '\n'
}

// Phase 1's decode of a whole 16-byte chunk (round 4: the kernel is bound by vector instructions, and the decode is nearly half of
// them -- 72 per chunk in the word-by-word form above, 53 here).  Same results, bit for bit:
//   code32  the chunk's packed 16-mer, first base in bits 31..30 (A0 C1 G2 T3; 0 for bytes that are no A/C/G/T)
//   inv16   bit i <=> byte i is not A/C/G/T (any case)
//   oddw    non-zero <=> some byte is neither A/C/G/T, N/n nor '\n' (a byte only the byte-string kernel can judge)
// What changed: (1) the four 2-bit codes of a word, and the four "invalid" flags of a word, are gathered by a byte dot product
// (v_dot4_u32_u8: one instruction where a multiply and a shift stood; the flags of two words accumulate through its addend);
// (2) the flag "this byte is not the letter its low three bits say it should be" is bit 7 of ((u ^ e) & 0x7F) + 0x7F OR'ed with
// bit 7 of the BYTE instead of bit 7 of u ^ e (one instruction less: where e = 0xFF the low seven bits already differ -- a byte
// whose low seven bits are all ones selects 'G' --, and where e is a letter its bit 7 is clear), with the ANDs and ORs folded into
// three-input bit operations (v_bitop3_b32); (3) the case fold is two instructions, the odd-byte word two per input word.
__device__ __forceinline__ void sk_decode16(const sk_u4 v, uint32_t &code32, uint32_t &inv16, uint32_t &oddw)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t r[4], f[4], odd = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t u   = __builtin_amdgcn_bitop3_b32(w[k], w[k] >> 1, 0x20202020u, 0x70);    // a & ~(b & c): letters upper-cased, everything else as it is
        const uint32_t sel = w[k] & 0x07070707u;
        const uint32_t cd  = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, sel);               // A0 C1 G2 T3, a byte each
        r[k] = __builtin_amdgcn_udot4(cd, 0x01041040u, 0u, false);                               // first base x 64 + ... + fourth x 1
        const uint32_t e   = __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, sel);               // the letter the byte would have to be (0xFF: none)
        const uint32_t t   = __builtin_amdgcn_bitop3_b32(u, e, 0x7F7F7F7Fu, 0x28) + 0x7F7F7F7Fu;   // (a ^ b) & c, + 0x7F: bit 7 <=> the low seven bits differ
        f[k] = __builtin_amdgcn_bitop3_b32(t, w[k], 0x80808080u, 0xA8);                          // (a | b) & c: 0x80 where the byte is no A/C/G/T
        const uint32_t px  = __builtin_amdgcn_perm(0x474EFF54u, 0x430A41FFu, sel);               // ... N and '\n' expected as well
        odd = __builtin_amdgcn_bitop3_b32(odd, px, u, 0xF6);                                      // a | (b ^ c)
    }
    code32 = (((((r[0] << 8) + r[1]) << 8) + r[2]) << 8) + r[3];
    const uint32_t lo = __builtin_amdgcn_udot4(f[1], 0x80402010u, __builtin_amdgcn_udot4(f[0], 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(f[3], 0x80402010u, __builtin_amdgcn_udot4(f[2], 0x08040201u, 0u, false), false);
    inv16 = ((hi << 8) + lo) >> 7;                                                                // (the flags are 0x80 each: everything x 128)
    oddw = odd;
}

// Among the (few) non-ACGT bytes of a 16-byte chunk, is there one that is neither N/n nor '\n'?
// Such a byte can only be judged by the exact byte-string kernel.  inv16 = the chunk's mask.
__device__ __forceinline__ uint32_t sk_chunk_has_odd_byte(const sk_u4 v, uint32_t inv16)
{
    uint32_t odd = 0;
    while (inv16) {
        const uint32_t i = (uint32_t)__builtin_ctz(inv16);
        inv16 &= inv16 - 1u;
        const uint32_t q = i >> 2;
        const uint32_t w = q == 0u ? v.x : q == 1u ? v.y : q == 2u ? v.z : v.w;
        const uint32_t b = (w >> (8u * (i & 3u))) & 0xFFu;
        odd |= (uint32_t)(((b & 0xDFu) != 'N') & (b != '\n'));
    }
    return odd;
}

#ifndef SK_PRIO_BASE
#define SK_PRIO_BASE 0                       // ... the decode section's priority
#endif
#ifndef SK_PRIO_P2
#define SK_PRIO_P2 SK_PRIO_BASE              // ... phase 2's (records read, hashes, the eight lookups issued)
#endif
#ifndef SK_PRIO_TAIL
#define SK_PRIO_TAIL SK_PRIO_P2              // ... everything behind that
#endif
#ifndef SK_PRIO_LVL
#define SK_PRIO_LVL 3
#endif
#ifndef SK_PRIO
#define SK_PRIO 1                            // wave priority (s_setprio): 1 = raised to SK_PRIO_LVL while a wave issues phase 1's stream loads, back to SK_PRIO_BASE
                                             // for the decode; 4 = kept up until the barrier.  A wave that starts a tile gets its nine loads out at once instead of
                                             // taking turns with the waves that decode: 0.724 -> 0.645 ms at cfg 2 (profiles/r03_kernel_experiments.txt, item 14)
#endif
#ifndef SK_SEED2_MIN
#define SK_SEED2_MIN 128                    // stage 2: in a wave with at least this many surviving chunks (of 512: a stretch of strain reads) a stretch without a seed
                                            // tries one more window before its windows go one by one; 0 = never, 1 = always.  Round 3 measured the compile-time
                                            // form: 1 % / 3 % divergence -5 % / -11 %, cfg 2 +1.4..2.7 % -- so the choice is made at run time, by the density the
                                            // wave already knows (round 4): cfg 2's waves (ten survivors) never try, a diverged genome's always do
#endif
#ifndef SK_RUN_PASS
#define SK_RUN_PASS 5                      // phase 2: this many level-1 survivors in a row (two more in a union table) go to stage 2 unquestioned;
                                           // 0 = never.  Measured (profiles/r03_kernel_experiments.txt, item 10): 3 costs 6.5 % with no strain reads (runs of three
                                           // false positives are frequent enough to send a wave in six down stage 2's slow path), 5 costs nothing there
                                           // and saves 3 % when every read is a strain read, 0.3 % at cfg 2
#endif
#ifndef SK_PHASE_CLOCK
#define SK_PHASE_CLOCK 0                    // experiment: wave-cycles per phase of the scan kernel, summed into the last words of the odd list (sk_debug_phase_clock)
#endif
#if SK_PHASE_CLOCK
#define SK_PHASE(k) do { if (!TALLY && !CAND) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        if (lane == 0u) atomicAdd((unsigned long long *)(table.oddlist + table.oddcap - 1024u) + (((blockIdx.x * SK_WAVES + (tid >> 6)) & 63u) * 8u + (k)), now_ - pc_last); \
        pc_last = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SK_PHASE(k) do { } while (0)
#endif
__device__ __forceinline__ uint32_t sk_revcomp32(uint32_t x)              // 16 packed bases
{
    uint32_t y = __builtin_bitreverse32(x);
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    return ~y;
}

// x = XOR of two packed 16-mers (first base in the top two bits): bit i of the result <=> base i differs
// (the bit order of the "not ACGT" masks)
__device__ __forceinline__ uint32_t sk_mismatch16(uint32_t x)
{
    uint32_t z = __builtin_bitreverse32((x | (x >> 1)) & 0x55555555u) >> 1;      // base i at bit 2i
    z = (z | (z >> 1)) & 0x33333333u;
    z = (z | (z >> 2)) & 0x0F0F0F0Fu;
    z = (z | (z >> 4)) & 0x00FF00FFu;
    return (z | (z >> 8)) & 0xFFFFu;
}

// both orientations of the window that ends at tile-relative position e, from the LDS records: packed 31-mers,
// first base in bits 61..60; the canonical form is the larger (src/genome_compare.c:1100-1120)
__device__ __forceinline__ void sk_window_keys(const uint32_t *rec, uint32_t e, uint64_t &fwd, uint64_t &rc)
{
    const uint32_t b = e + SK_SPAN;                        // record 0 holds the 128 bases before the tile
    const uint32_t c = b >> 4, s = 2u * (15u - (b & 15u));
    const uint32_t w0 = rec[(c >> 3) * SK_REC_DW + (c & 7u)];
    const uint32_t w1 = rec[((c - 1u) >> 3) * SK_REC_DW + ((c - 1u) & 7u)];
    const uint32_t w2 = rec[((c - 2u) >> 3) * SK_REC_DW + ((c - 2u) & 7u)];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, s);
    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, s) & 0x3FFFFFFFu;
    fwd = ((uint64_t)hi << 32) | lo;
    uint64_t r = ((uint64_t)__builtin_bitreverse32(lo) << 32) | __builtin_bitreverse32(hi);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    rc = (~r) >> 2;
}

// the 31 bases of the strain's text that start at position q, packed like a key (q + 31 <= text_bases; the
// text array has two spare words behind its last base)
__device__ __forceinline__ uint64_t sk_text_key(const uint32_t *__restrict__ text2, uint32_t q)
{
    const uint32_t w = q >> 4, o2 = 2u * (q & 15u);
    const uint32_t t0 = text2[w], t1 = text2[w + 1u], t2 = text2[w + 2u];
    const uint64_t x = ((uint64_t)t0 << 32) | t1;
    return ((x << o2) | ((uint64_t)t2 >> (32u - o2))) >> 2;
}

__device__ __forceinline__ bool sk_grid_test(const uint2 blk, uint32_t bits)
{
    // the four bit positions are the low five bits of the four BYTES of `bits`: a shift takes its amount straight from a byte of
    // a register (SDWA), so a position costs no instruction of its own (round 3; same false-positive rate as the 5-bit fields it replaces)
    const uint32_t t = (blk.x >> ((bits >> 24) & 31u)) & (blk.x >> ((bits >> 16) & 31u)) &
                       (blk.y >> ((bits >> 8) & 31u)) & (blk.y >> (bits & 31u));
    return (t & 1u) != 0u;
}

// Level 2 is keyed on 24-MERS (round 3).  Every window of a chunk holds, whole, either the chunk and the 8 bases before it or the
// chunk and the 8 bases behind it (windows that begin 8..15 bases before the chunk: the first; the others: the second) -- so a
// chunk neither of whose two 24-mers is a 24-mer of the strain has no window left.  A 16-mer stops being selective when the table
// is a union of many strains (32 x 5 Mbp: 7.5 % of ALL canonical 16-mers are in it, every one of them a true positive of a filter
// keyed on 16-mers, each then costing a table probe and sixteen windows one by one); a 24-mer of unrelated DNA is in no table.
// f: 24 packed bases, the first in bits 47..46; canonical = the smaller of the two orientations; two hashes of the 48 bits, one for
// the block and one for the bits in it (with a single 32-bit hash the union's 160 M entries would collide with 4 % of all questions).
__device__ __forceinline__ uint64_t sk_canon24(uint64_t f)
{
    uint64_t y = __builtin_bitreverse64(f) >> 16;
    y = ((y >> 1) & 0x555555555555ull) | ((y & 0x555555555555ull) << 1);
    const uint64_t rc = ~y & 0xFFFFFFFFFFFFull;
    return f < rc ? f : rc;
}
__device__ __forceinline__ uint32_t sk_h24_block(uint64_t c24) { return sk_gmix((uint32_t)c24 ^ ((uint32_t)(c24 >> 32) * 0x9E3779B1u)); }
__device__ __forceinline__ uint32_t sk_h24_bits(uint64_t c24) { return sk_gmix((uint32_t)c24 * 0x7FEB352Du + (uint32_t)(c24 >> 32) * 0x846CA68Bu); }
// the 24 bases of the text that start at position q (the text array has two spare words behind its last base)
__device__ __forceinline__ uint64_t sk_text_24(const uint32_t *__restrict__ text2, uint32_t q)
{
    const uint32_t w = q >> 4, o2 = 2u * (q & 15u);
    const uint64_t x = ((uint64_t)text2[w] << 32) | text2[w + 1u];
    return ((x << o2) | ((uint64_t)text2[w + 2u] >> (32u - o2))) >> 16;
}
__device__ __forceinline__ void sk_grid2_insert24(uint32_t *__restrict__ w2, uint32_t shift2, uint64_t f24)
{
    const uint64_t c24 = sk_canon24(f24);
    const uint32_t b = sk_grid2_bits(sk_h24_bits(c24));
    uint32_t *blk = w2 + 2u * (size_t)sk_grid2_block(sk_h24_block(c24), shift2);
    const uint32_t m0 = (1u << ((b >> 24) & 31u)) | (1u << ((b >> 16) & 31u)), m1 = (1u << ((b >> 8) & 31u)) | (1u << (b & 31u));
    if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
    if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
}

// level-1 filter verdict on one packed 16-mer (either orientation): false = certainly not in the strain
__device__ __forceinline__ bool sk_grid1_has(const sk_table_view &t, uint32_t w16)
{
    const uint32_t r = sk_revcomp32(w16);
    const uint32_t g = sk_gmix(w16 < r ? w16 : r);
    const uint2 q = t.grid1[sk_grid1_block(g, t.grid1_blocks)];
    return sk_grid_test(q, sk_grid1_bits(g));
}

// ---------------------------------------------------------------------------------------------
// THE hot kernel, second generation ("grid"): no per-base work after the decode.
//
// The stream is cut into 16-base chunks at multiples of 16.  A 31-base window contains exactly one
// whole chunk, so the windows are partitioned by chunk, 16 each (the windows ENDING at chunk start
// + 15 .. + 30).  A workgroup owns the SK_TILE/16 chunks of its tile and all their windows.
//
//   phase 1  as before: 16 stream bytes -> one packed code word + a 16-bit "not ACGT" mask in LDS.
//            A chunk's code word IS its packed 16-mer.
//   phase 2  per chunk (8 per thread, their filter loads issued together): canonical 16-mer
//            (min of the word and its reverse complement), one 8-byte load from the L2-resident
//            level-1 filter "is this 16-mer in the strain at all, in either orientation?".  Reads
//            unrelated to the strain stop here (~5 % false positives), having cost ~1.5 VALU
//            operations per base.  Survivors ask the large level-2 filter (false positives ~1e-5).
//   stage 2  the live windows of the surviving chunks (<= 16 each; liveness from the masks of the
//            chunk's two neighbours) are queued as tile positions, one thread's chunks after the
//            other so that consecutive windows sit in consecutive queue slots, and probed 64 at a
//            time exactly as in sk_scan_main: anchors through the hash, followers through their
//            anchor's neighbour in strain order, every hit a full 62-bit compare.
// ---------------------------------------------------------------------------------------------
// invalid mask of chunk c (index into the LDS records: record 0 = the 8 chunks before the tile)
__device__ __forceinline__ uint32_t sk_chunk_inv(const uint32_t *rec, uint32_t c)
{
    return ((const uint16_t *)rec)[(c >> 3) * (2 * SK_REC_DW) + 16 + (c & 7u)];
}

// the read bases are touched once: keep them from pushing the filter out of the L2
__device__ __forceinline__ sk_u4 sk_stream_load(const sk_u4 *p)
{
#if SK_STREAM_POLICY == 0
    return __builtin_nontemporal_load(p);
#elif SK_STREAM_POLICY == 1
    return *p;
#else
    sk_u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
#endif
}

// the first `keep` (< 16) bytes of a chunk, '\n' behind them
__device__ __forceinline__ sk_u4 sk_mask_tail(sk_u4 v, uint32_t keep)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = (int)keep - 4 * j;                           // bytes of this word that are kept
        const uint32_t m = k >= 4 ? 0xFFFFFFFFu : k <= 0 ? 0u : (1u << (8 * k)) - 1u;
        w[j] = (w[j] & m) | (0x0A0A0A0Au & ~m);
    }
    return (sk_u4){w[0], w[1], w[2], w[3]};
}

// one 16-byte chunk of the stream at byte offset off (may start before 0 or end beyond nbytes: '\n' fill there)
__device__ __forceinline__ sk_u4 sk_load_chunk(const uint8_t *__restrict__ stream, uint64_t nbytes, int64_t off)
{
    if (off >= 0 && (uint64_t)off + 16u <= nbytes) return sk_stream_load((const sk_u4 *)(stream + off));
    uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
    for (int i = 0; i < 16; i++) {
        const int64_t p = off + i;
        if (p >= 0 && (uint64_t)p < nbytes)
            w[i >> 2] = (w[i >> 2] & ~(0xFFu << ((i & 3) * 8))) | ((uint32_t)stream[p] << ((i & 3) * 8));
    }
    return (sk_u4){w[0], w[1], w[2], w[3]};
}

// CAND: third pass of the partitioned pipeline (sk_bin -> sk_lds_probe -> this): `cand` holds one byte per chunk of the
// batch, non-zero for the chunks the LDS-resident filter slices could not rule out; only those chunks (and the lines
// around them) are read and looked at.
// Scalar registers decide how many workgroups a CU admits: <= 80 -> 8 of these 256-thread groups, 81..96 -> 7, 97..112 -> 6
// (MI355X_MICROARCH.md, "Residency").  The COUNT kernel is held at 80, the TALLY kernels at 96 (round 3: 100 and 94 = 6 and 7 groups).
template <bool TALLY, int ABLATE, bool CAND, bool UNION = false, bool PACKED = false>
#if defined(SK_NO_SGPR_CAP)                                       // (A/B builds: round 3's register budget -- TALLY 100 scalar registers = 6 groups per CU, UNION 94 = 7)
__global__ __launch_bounds__(SK_THREADS)
#else
__global__ __launch_bounds__(SK_THREADS) __attribute__((amdgpu_num_sgpr(80)))
#endif
void sk_scan_grid(const uint8_t *__restrict__ stream, uint64_t nbytes, uint64_t emit_begin,
                  sk_table_view table, sk_sink sink, uint32_t *__restrict__ flags, const uint8_t *__restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) uint32_t rec[(SK_NREC + 1) * SK_REC_DW];
    __shared__ __attribute__((aligned(16))) uint16_t wq_all[SK_WAVES][128 + 16];   // below 128 before a push of at most 16 (phase 2 borrows its first 64 bytes)
    __shared__ uint16_t cq_all[SK_WAVES][64 * SK_SPAN_CH];       // the wave's surviving chunks
    // COUNT mode: difference-array indices this workgroup has already touched once in this tile; further updates
    // of them are added up here and flushed with one atomic each at the end.  Reads that repeat (duplicates)
    // would otherwise serialise on a few words in the L2 (same-address atomics).
    __shared__ uint2 agg[SK_AGG];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
#if SK_PHASE_CLOCK
    unsigned long long pc_last = __builtin_amdgcn_s_memtime();
#endif
    __shared__ uint32_t hl_n[2];                                  // TALLY: the hit log's LDS share (agg is free in that mode)
    if (!TALLY)
        for (uint32_t i = tid; i < SK_AGG; i += SK_THREADS) agg[i] = make_uint2(0xFFFFFFFFu, 0u);   // (visible after phase 1's barrier)
    else {
        if (tid == 0u) { hl_n[0] = 0u; hl_n[1] = SK_AGG; }
        sink.lds_hits = agg;
        sink.lds_n = hl_n;
    }
    const uint64_t tile0 = (uint64_t)blockIdx.x * SK_TILE;        // stream offset of the tile's first chunk
    uint32_t bad = 0;

    // A tile's stream loads: all of a thread's 16-byte loads are issued together (nine HBM latencies in flight instead of one
    // after the other), at raised wave priority -- a wave that starts a tile must not take turns, instruction by instruction, with
    // the waves that decode or hash (round 3).  Tiles at the ends of the batch mask what lies outside it.
    constexpr int NIT = CAND ? 1 : (SK_NCHUNK_GRID + SK_THREADS - 1) / SK_THREADS;
    sk_u4 vv[NIT];
    auto issue_loads = [&](uint64_t t0) {
        const bool inside = t0 >= SK_SPAN && t0 + SK_TILE + 16u <= nbytes;      // (workgroup-uniform)
#if SK_PRIO & 1
        __builtin_amdgcn_s_setprio(SK_PRIO_LVL);
#endif
        if (inside) {
    #pragma unroll
            for (int it = 0; it < NIT; it++) {
                const uint32_t c = tid + (uint32_t)it * SK_THREADS;
                vv[it] = (sk_u4){0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
                if (c < SK_NCHUNK_GRID) vv[it] = sk_stream_load((const sk_u4 *)(stream + (t0 - SK_SPAN) + (uint64_t)c * 16u));
            }
        } else {
            // a tile at either end of the batch: chunks that lie outside it read as separators, and so do the bytes of the last chunk
            // beyond the batch's end.  The chunk itself is loaded whole -- the stream is 16-byte aligned, so a chunk that begins
            // inside the batch lies in a mapped page to its last byte
    #pragma unroll
            for (int it = 0; it < NIT; it++) {
                const uint32_t c = tid + (uint32_t)it * SK_THREADS;
                const int64_t off = (int64_t)t0 - SK_SPAN + (int64_t)c * 16;
                sk_u4 v = (sk_u4){0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};       // '\n' fill
                if (c < SK_NCHUNK_GRID && off >= 0 && (uint64_t)off < nbytes) {
                    v = sk_stream_load((const sk_u4 *)(stream + off));
                    const uint64_t rem = nbytes - (uint64_t)off;
                    if (rem < 16u) v = sk_mask_tail(v, (uint32_t)rem);
                }
                vv[it] = v;
            }
        }
#if (SK_PRIO & 5) == 1
        __builtin_amdgcn_s_setprio(SK_PRIO_BASE);
#endif
    };
    if (!CAND && !PACKED) issue_loads(tile0);

    // ================= phase 1: bytes -> packed codes + invalid masks ==========================
    uint32_t candm = 0xFFu;                                        // this thread's chunks that are candidates (CAND)
    if (PACKED) {
        // The batch came PACKED by the host (sk_pack_stream: what sk_decode16 makes of a 16-byte chunk, made there -- 6 bytes a chunk
        // over the PCIe link instead of 16): `stream` holds the chunks' code words, `cand` their masks of bytes that are no A/C/G/T.
        // A packed batch holds no byte the byte-string kernel would have to judge (the host sends such a batch as bytes).  Chunks
        // outside the batch read as separators, as in the byte form.
        const uint32_t *__restrict__ codes = (const uint32_t *)stream;
        const uint16_t *__restrict__ invs = (const uint16_t *)cand;
        const int64_t g0 = (int64_t)(tile0 >> 4) - (int64_t)SK_SPAN_CH;
        const int64_t nch = (int64_t)((nbytes + 15u) >> 4);
        uint32_t pc[NIT], pi[NIT];
    #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t c = tid + (uint32_t)it * SK_THREADS;
            const int64_t g = g0 + (int64_t)c;
            pc[it] = 0u; pi[it] = 0xFFFFu;
            if (c < SK_NCHUNK_GRID && g >= 0 && g < nch) { pc[it] = __builtin_nontemporal_load(codes + g); pi[it] = __builtin_nontemporal_load(invs + g); }
        }
    #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t c = tid + (uint32_t)it * SK_THREADS;
            if (c < SK_NCHUNK_GRID) {
                const uint32_t r = c >> 3, sl = c & 7u;
                rec[r * SK_REC_DW + sl] = pc[it];
                ((uint16_t *)rec)[r * (2 * SK_REC_DW) + 16 + sl] = (uint16_t)pi[it];
            }
        }
    } else
    if (!CAND) {
        SK_PHASE(0);                                                 // start -> the tile's loads issued
    #pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t c = tid + (uint32_t)it * SK_THREADS;
            if (c < SK_NCHUNK_GRID) {
                const sk_u4 v = vv[it];
                uint32_t code32, inv16, oddw;
                sk_decode16(v, code32, inv16, oddw);
                const bool odd = oddw != 0u;
                // bytes of the chunk after the tile belong to the next tile, which reports them itself
                if (c < SK_NCHUNK && odd) {
                    bad = 1;
                    if (c >= SK_SPAN_CH) {                                   // (the chunks before the tile are the previous tile's)
                        const uint32_t at = atomicAdd(&flags[2], 1u);
                        if (at < table.oddcap) table.oddlist[at] = (uint32_t)((tile0 - SK_SPAN + (uint64_t)c * 16u) >> 4);
                    }
                }
                const uint32_t r = c >> 3, sl = c & 7u;
                rec[r * SK_REC_DW + sl] = code32;
                ((uint16_t *)rec)[r * (2 * SK_REC_DW) + 16 + sl] = (uint16_t)inv16;
            }
        }
    } else {
        // the candidate bytes of this thread's eight chunks and of the two chunks next to them
        const uint64_t nch = (nbytes + 15u) >> 4, c8 = (tile0 >> 4) + (uint64_t)tid * SK_SPAN_CH;
        uint64_t own = 0ull;
        if (c8 + SK_SPAN_CH <= nch) own = *(const uint64_t *)(cand + c8);
        else for (uint32_t i = 0; i < SK_SPAN_CH; i++) if (c8 + i < nch) own |= (uint64_t)cand[c8 + i] << (8u * i);
        own = (own | (own >> 1) | (own >> 2) | (own >> 4)) & 0x0101010101010101ull;          // (any non-zero byte value of 1..0x17)
        candm = (uint32_t)((own * 0x0102040810204080ull) >> 56);
        const uint32_t before = c8 > 0 && c8 - 1u < nch ? cand[c8 - 1u] : 0u, after = c8 + SK_SPAN_CH < nch ? cand[c8 + SK_SPAN_CH] : 0u;
        if (ABLATE == 7) { if (own == 0x123456789ull && before + after == 77u) flags[3] = 1u; return; }    // timing: the candidate map alone
        // only the chunks next to a candidate are read (a candidate needs itself and its two neighbours)
        const int64_t off0 = (int64_t)tile0 + (int64_t)tid * SK_SPAN;
        const uint32_t needm = (candm | (candm << 1) | (candm >> 1) | (before ? 1u : 0u) | (after ? 0x80u : 0u)) & 0xFFu;
        if (needm) {
            sk_u4 vv[SK_SPAN_CH];
            if (tile0 + SK_TILE + 16u <= nbytes) {                 // (workgroup-uniform) plain predicated loads, all in flight together
#pragma unroll
                for (int i = 0; i < SK_SPAN_CH; i++) {
                    vv[i] = (sk_u4){0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};
                    if ((needm >> i) & 1u) vv[i] = sk_stream_load((const sk_u4 *)(stream + off0) + i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < SK_SPAN_CH; i++)
                    if ((needm >> i) & 1u) vv[i] = sk_load_chunk(stream, nbytes, off0 + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < SK_SPAN_CH; i++) {
                if (!((needm >> i) & 1u)) continue;
                uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
                sk_decode4(vv[i].x, c0, i0);
                sk_decode4(vv[i].y, c1, i1);
                sk_decode4(vv[i].z, c2, i2);
                sk_decode4(vv[i].w, c3, i3);
                rec[(tid + 1u) * SK_REC_DW + i] = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
                ((uint16_t *)rec)[(tid + 1u) * (2 * SK_REC_DW) + 16 + i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
            }
        }
        // the chunk before the tile and the one after it (other tiles' chunks, needed next to a candidate at the edge)
        if ((tid == 0u && (candm & 1u)) || (tid == SK_THREADS - 1u && (candm >> 7))) {
            const bool lead = tid == 0u;
            const sk_u4 v = sk_load_chunk(stream, nbytes, lead ? (int64_t)tile0 - 16 : (int64_t)tile0 + SK_TILE);
            uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
            sk_decode4(v.x, c0, i0);
            sk_decode4(v.y, c1, i1);
            sk_decode4(v.z, c2, i2);
            sk_decode4(v.w, c3, i3);
            const uint32_t r = lead ? 0u : SK_NREC, sl = lead ? 7u : 0u;
            rec[r * SK_REC_DW + sl] = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
            ((uint16_t *)rec)[r * (2 * SK_REC_DW) + 16 + sl] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
        }
    }
    SK_PHASE(1);                                                     // waiting for the loads + decode
#if (SK_PRIO & 5) == 5
    __builtin_amdgcn_s_setprio(SK_PRIO_BASE);
#endif
    __syncthreads();
    SK_PHASE(2);                                                     // the barrier
    if (ABLATE == 8) { if (rec[tid] == 0x12345u && candm == 0x77u) flags[3] = 1u; return; }               // timing: phase 1 alone

    // ================= phase 2: one filter lookup per chunk ======================================
#if SK_PRIO_P2 != SK_PRIO_BASE
    __builtin_amdgcn_s_setprio(SK_PRIO_P2);
#endif
    uint16_t *const wq = wq_all[tid >> 6];
    uint16_t *const cq = cq_all[tid >> 6];                        // the wave's list of chunks: phase 2's questions first, stage 2's survivors then
    uint32_t qw = 0;                                              // queue fill (wave-uniform)
    const uint32_t *my = rec + (tid + 1u) * SK_REC_DW;            // this thread's 8 chunks

    uint32_t g[SK_SPAN_CH];
    uint2    b1[SK_SPAN_CH];
    uint32_t okm = 0;                                             // chunks without a non-ACGT byte
    // the thread's record in three 16-byte reads: with the 12-dword lane stride these are conflict-free, the same
    // words read one by one are four-way bank conflicts (12 x 8 lanes = 96 = 0 mod 32 banks)
    const sk_u4 rq0 = *(const sk_u4 *)my, rq1 = *(const sk_u4 *)(my + 4), rq2 = *(const sk_u4 *)(my + 8);
    const uint32_t rw[12] = {rq0.x, rq0.y, rq0.z, rq0.w, rq1.x, rq1.y, rq1.z, rq1.w, rq2.x, rq2.y, rq2.z, rq2.w};
#pragma unroll
    for (int i = 0; i < SK_SPAN_CH; i++) {
        const uint32_t cw = rw[i];
        const uint32_t ipair = rw[8 + (i >> 1)];
        const uint32_t inv = (i & 1) ? ipair >> 16 : ipair & 0xFFFFu;
        const uint32_t rc = sk_revcomp32(cw);
        g[i] = sk_gmix(cw < rc ? cw : rc);
        okm |= (uint32_t)(inv == 0u && (!CAND || ((candm >> i) & 1u))) << i;
        b1[i] = make_uint2(0u, 0u);
        if (ABLATE == 4) { if (inv == 0u) b1[i] = table.grid1[sk_grid1_block(g[i], table.grid1_blocks) & 131071u]; }  // timing: all lookups in 1 MiB (L2 hits)
        else if (ABLATE == 6) { if (inv == 0u) b1[i] = table.grid1[sk_grid1_block(g[i], table.grid1_blocks) & 2047u]; } // timing: all lookups in 16 KiB (L1 hits)
        else if (ABLATE == 10) { if (inv == 0u) b1[i] = table.grid1[sk_grid1_block(g[i], table.grid1_blocks)]; }         // timing: the real level-1 lookups, verdicts dropped (no level 2, no stage 2)
        else if (ABLATE != 1 && ((okm >> i) & 1u)) b1[i] = table.grid1[sk_grid1_block(g[i], table.grid1_blocks)];
    }
    SK_PHASE(3);                                                     // records read, hashes, lookups issued
#if SK_PRIO_TAIL != SK_PRIO_P2
    __builtin_amdgcn_s_setprio(SK_PRIO_TAIL);
#endif
    uint32_t m = 0;                                               // chunks that may be in the strain
#pragma unroll
    for (int i = 0; i < SK_SPAN_CH; i++)
        m |= (uint32_t)(((okm >> i) & 1u) != 0u && sk_grid_test(b1[i], sk_grid1_bits(g[i])) &&
                        ((ABLATE != 4 && ABLATE != 6 && ABLATE != 10) || g[i] == 0x9E3779B9u)) << i;     // (ablations: loads kept alive, verdicts dropped)
    SK_PHASE(4);                                                     // waiting for the lookups + their verdicts
#if SK_RUN_PASS
    // SK_RUN_PASS level-1 survivors in a row (two more in a union table, where one chunk in seven passes level 1 by chance) are a read of
    // the strain: the run goes to stage 2 unquestioned, without the round trips to the L2 and to HBM that the questions below cost a
    // wave whose strain read waits for them.  Pruning less is always exact.  The neighbouring lanes' verdicts carry a run over the
    // edge of a thread's eight chunks (a 150-base read is nine chunks: with own chunks only, one of its two threads still asks).
    uint32_t runpass = 0u;
    if (!CAND && ABLATE != 5) {
        constexpr uint32_t RUN = UNION ? SK_RUN_PASS + 2u : SK_RUN_PASS, MARGIN = RUN - 1u;
        static_assert(MARGIN <= 8u, "a run longer than nine chunks needs more than the two neighbouring lanes");
        const uint32_t mu = (uint32_t)__shfl_up((int)m, 1), md = (uint32_t)__shfl_down((int)m, 1);
        const uint32_t wide = (lane > 0u ? mu >> (8u - MARGIN) : 0u) | (m << MARGIN) | (lane < 63u ? (md & ((1u << MARGIN) - 1u)) << (8u + MARGIN) : 0u);
        uint32_t sr = wide, members = 0u;                                                                       // chunks -MARGIN .. 7 + MARGIN of this thread
#pragma unroll
        for (uint32_t k = 1; k < RUN; k++) sr &= wide >> k;                                                     // a run of RUN starts here
#pragma unroll
        for (uint32_t k = 0; k < RUN; k++) members |= sr << k;
        runpass = (members >> MARGIN) & m;
    }
#endif
    // Level 2, a CHUNK PER LANE (round 3, when the kernel had become bound by vector instructions).  The false positives of level 1 are
    // 7 % of the chunks: nearly every thread-wise loop over "my survivors" runs in every wave (some lane always has one), one or two
    // rounds of ~220 instructions for two or three busy lanes.  Here the chunks to be asked (level-1 survivors outside the runs that pass
    // unquestioned) are compacted over the wave -- 36 of 512 on average: ONE round with half the lanes busy -- each lane asks about one
    // chunk (the two half-shifted 16-mers, then the 24-mer of a side that passed), and the verdicts go back to the owners as bits of a
    // word in LDS.  Every asked chunk stands for itself (no "right behind a chunk that passed"): strain reads are the runs.
    if (ABLATE != 5 && !CAND) {
        uint32_t ask = m;
#if SK_RUN_PASS
        ask &= ~runpass;
#endif
        const unsigned long long anyask = __ballot(ask != 0u);
        uint32_t m2 = m & ~ask;
        if (anyask) {                                             // (wave-uniform)
            uint32_t *const pb = (uint32_t *)wq;                  // 64 bytes: one verdict byte per lane of the wave
            if (lane < 16u) pb[lane] = 0u;
            uint32_t incl = (uint32_t)__popc(ask);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            const uint32_t nask = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            {
                uint32_t at = incl - (uint32_t)__popc(ask), a = ask;
                while (a) { const uint32_t i = (uint32_t)__builtin_ctz(a); a &= a - 1u; cq[at++] = (uint16_t)(tid * SK_SPAN_CH + i); }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b0 = 0; b0 < nask; b0 += 64u) {         // (wave-uniform)
                if (b0 + lane < nask) {
                    const uint32_t ct = cq[b0 + lane];                                       // chunk of the tile: owner thread * 8 + i
                    const uint32_t cid = ct + SK_SPAN_CH;                                    // record 0 = the 8 chunks before the tile
                    const uint32_t cw  = rec[(cid >> 3) * SK_REC_DW + (cid & 7u)];
                    const uint32_t cwp = rec[((cid - 1u) >> 3) * SK_REC_DW + ((cid - 1u) & 7u)];
                    const uint32_t cwn = rec[((cid + 1u) >> 3) * SK_REC_DW + ((cid + 1u) & 7u)];
                    const uint32_t ivp = sk_chunk_inv(rec, cid - 1u), ivn = sk_chunk_inv(rec, cid + 1u);
                    const bool lv_ok = (ivp >> 8) == 0u, rv_ok = (ivn & 0xFFu) == 0u;        // the 8 bases borrowed from either neighbour are ACGT
                    const uint32_t wl = __builtin_amdgcn_alignbit(cwp, cw, 16), wr = __builtin_amdgcn_alignbit(cw, cwn, 16);
                    const uint32_t rl = sk_revcomp32(wl), rr = sk_revcomp32(wr);
                    const uint32_t gl = sk_gmix(wl < rl ? wl : rl), gr = sk_gmix(wr < rr ? wr : rr);
                    uint2 bl = make_uint2(0u, 0u), br = make_uint2(0u, 0u);                  // (both lookups in flight together)
                    if (lv_ok) bl = table.grid1[sk_grid1_block(gl, table.grid1_blocks)];
                    if (rv_ok) br = table.grid1[sk_grid1_block(gr, table.grid1_blocks)];
                    const bool al = lv_ok && sk_grid_test(bl, sk_grid1_bits(gl)), ar = rv_ok && sk_grid_test(br, sk_grid1_bits(gr));
                    bool pass = false;
                    if (al | ar) {                                // level 2: the 24-mer of a side that passed, the other side's only if that one is no 24-mer of the strain
                        const uint64_t l24 = ((uint64_t)(cwp & 0xFFFFu) << 32) | cw, r24 = ((uint64_t)cw << 16) | (cwn >> 16);
                        uint64_t c24 = sk_canon24(al ? l24 : r24);
                        pass = sk_grid_test(table.grid2[sk_grid2_block(sk_h24_block(c24), table.grid2_shift)], sk_grid2_bits(sk_h24_bits(c24)));
                        if (!pass && al && ar) {
                            c24 = sk_canon24(r24);
                            pass = sk_grid_test(table.grid2[sk_grid2_block(sk_h24_block(c24), table.grid2_shift)], sk_grid2_bits(sk_h24_bits(c24)));
                        }
                    }
                    if (pass) atomicOr(&pb[(ct >> 5) & 15u], 1u << (ct & 31u));              // owner lane = (ct >> 3) & 63: byte (lane & 3) of word lane >> 2
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            m2 |= (pb[lane >> 2] >> ((lane & 3u) * 8u)) & 0xFFu;
            __builtin_amdgcn_wave_barrier();                      // (wq is stage 2's again)
        }
        m = m2;
    }

    SK_PHASE(5);                                                     // the second and third questions
    if (ABLATE == 9) { if (m == 0x77u && tid == 100u) flags[3] = 1u; return; }                            // timing: phases 1 and 2 alone

    // ================= stage 2: the windows of the surviving chunks ==============================
    // Seed and verify, a CHUNK per lane (16 windows at a time, 64 chunks = up to 1024 windows per round).
    //   seed    a few chunks of every stretch of consecutive surviving chunks (its first, then every
    //           SK_ANCHOR_CH-th) probe ONE window in the HBM table; a hit comes back with the row's place in the
    //           strain's text and its orientation: the diagonal the read lies on.
    //   verify  every chunk of the stretch takes the nearest seed and compares its 48 bases (the chunk and its
    //           two neighbours: all that its 16 windows cover) with the strain's 2-bit text on that diagonal,
    //           three 32-bit XORs; mismatching and non-ACGT bases together give the verified windows by the same
    //           31-run bit trick that gives the live ones.  A verified window IS the text's k-mer at its place
    //           (all 62 bits were compared), and its row is the rank of that place (rank map: 16 bytes per 64
    //           positions) -- unless that place is a repeat of an earlier k-mer (no bit there), which goes to
    //   count   COUNT mode: verified windows with consecutive rows are consecutive counters, so a run of them is
    //           "+1 at its first row, -1 behind its last" in a difference array that the host side folds into the
    //           column by a prefix sum before anyone looks (sk_diff_flush); runs that continue in the next chunk
    //           cancel their inner ends.  Two atomics per read and strand instead of one per window.
    //           TALLY mode: the verified windows are spread over the lanes again, one each, for the per-read tallies.
    //   rest    windows the diagonal does not explain (a read error, a repeat, no seed hit) are queued one by one,
    //           asked about in the L2-resident level-1 filter (first and last 16-mer: they cover all 31 bases)
    //           and only then probed in the table.
    auto count_row = [&](uint32_t row, uint32_t pos) { sk_on_hit<false, ABLATE == 3>(sink, row, pos); };
    // difference-array update through the workgroup's table of indices already touched in this tile
    auto diff_add = [&](uint32_t idx, uint32_t delta) {
        const uint32_t a = (idx * 0x9E3779B1u) >> (32 - SK_AGG_LOG2);
        const uint32_t old = atomicCAS(&agg[a].x, 0xFFFFFFFFu, idx);
        if (old == idx) atomicAdd(&agg[a].y, delta);
        else if (ABLATE != 3) atomicAdd(&sink.diff[idx], delta);
    };

    auto probe_windows = [&](uint32_t n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        qw -= n;
        bool act[2];
        uint32_t e[2], hit[2];
        uint64_t cn[2];
        uint2 q0[2], q1[2];
        uint32_t g0[2], g1[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const uint32_t idx = lane + 64u * (uint32_t)s2;
            act[s2] = idx < n;
            e[s2] = act[s2] ? wq[qw + idx] : 0u;
            uint64_t fwd = 0ull, rc = 0ull;
            if (act[s2]) sk_window_keys(rec, e[s2], fwd, rc);
            cn[s2] = fwd > rc ? fwd : rc;
            hit[s2] = 0xFFFFFFFFu;
            const uint32_t w0 = (uint32_t)(cn[s2] >> 30), w1 = (uint32_t)cn[s2];
            const uint32_t r0 = sk_revcomp32(w0), r1 = sk_revcomp32(w1);
            g0[s2] = sk_gmix(w0 < r0 ? w0 : r0); g1[s2] = sk_gmix(w1 < r1 ? w1 : r1);
            q0[s2] = q1[s2] = make_uint2(0u, 0u);
            if (act[s2]) { q0[s2] = table.grid1[sk_grid1_block(g0[s2], table.grid1_blocks)]; q1[s2] = table.grid1[sk_grid1_block(g1[s2], table.grid1_blocks)]; }
        }
        if (ABLATE == 2) { __builtin_amdgcn_wave_barrier(); return; }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
            if (act[s2] && sk_grid_test(q0[s2], sk_grid1_bits(g0[s2])) && sk_grid_test(q1[s2], sk_grid1_bits(g1[s2]))) {
                uint32_t unused;
                hit[s2] = sk_find(cn[s2], table, &unused);
            }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            if (TALLY) sk_tally_wave<UNION>(sink, hit[s2], (uint32_t)tile0 + e[s2], lane);
            else if (hit[s2] != 0xFFFFFFFFu) count_row(hit[s2], (uint32_t)tile0 + e[s2]);
        }
        __builtin_amdgcn_wave_barrier();
    };

    // the wave's surviving chunks (index in the tile), compacted in stream order
    uint32_t nq = 0;
    if (__ballot(m != 0u) != 0ull) {                              // (wave-uniform; most waves of an unrelated metagenome have none)
        uint32_t incl = (uint32_t)__popc(m);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        nq = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t at = incl - (uint32_t)__popc(m);
#pragma unroll
        for (int i = 0; i < SK_SPAN_CH; i++)
            if ((m >> i) & 1u) cq[at++] = (uint16_t)(tid * SK_SPAN_CH + (uint32_t)i);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    for (uint32_t b0 = 0; b0 < nq; b0 += 64u) {                   // (wave-uniform)
        const bool act = b0 + lane < nq;
        const uint32_t ch = act ? cq[b0 + lane] : 0x7FFF0000u + 2u * lane;     // inactive: never consecutive
        // live windows of the chunk: bit j <=> the 31 bases ending at chunk start + 15 + j are all ACGT (the chunk
        // itself is clean: the masks of its two neighbours decide), and the window's end is not before emit_begin
        uint32_t live = 0u, inv_prev = 0u, inv_next = 0u;
        if (act) {
            inv_prev = sk_chunk_inv(rec, ch + SK_SPAN_CH - 1u);
            inv_next = sk_chunk_inv(rec, ch + SK_SPAN_CH + 1u);
            const uint64_t v = ~((uint64_t)inv_prev | ((uint64_t)inv_next << 32)) & 0x0000FFFFFFFFFFFFull;
            uint64_t rr = v & (v << 1);
            rr &= rr << 2;
            rr &= rr << 4;
            rr &= rr << 8;
            rr &= rr << 15;                                        // runs of >= 31
            live = (uint32_t)(rr >> 31) & 0xFFFFu;
            const uint64_t p0 = tile0 + ch * 16u + 15u;            // END of the chunk's first window
            if (p0 < emit_begin) {
                const uint64_t dlt = emit_begin - p0;
                live = dlt >= 16u ? 0u : live & (0xFFFFu << (uint32_t)dlt);
            }
        }
        uint32_t fb = live;                                       // windows left to the one-by-one path
        if (table.text2 != nullptr) {
            const uint32_t ch_prev = (uint32_t)__shfl_up((int)ch, 1);
            const bool first = (lane == 0u) | (ch != ch_prev + 1u);
            const unsigned long long fm = __ballot(first);
            const uint32_t lo_lane = 63u - (uint32_t)__clzll(fm & (~0ull >> (63u - lane)));
            const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
            const uint32_t hi_lane = above ? (uint32_t)__builtin_ctzll(above) - 1u : 63u;
            const uint32_t kk = lane - lo_lane;
            // ---- seed: one window of the chunk (its last live one: it reaches furthest into the read) in the table
            bool anchor = act & (live != 0u) & ((kk & (SK_ANCHOR_CH - 1u)) == 0u), use_first = false;
            uint32_t a_dir = 0u, a_diag = 0u;
            bool seed = false;
#pragma nounroll
            for (int pass = 0; pass < 2; pass++) {
                if (anchor) {
                    const uint32_t j = use_first ? (uint32_t)__builtin_ctz(live) : 31u - (uint32_t)__clz((int)live);
                    const uint32_t e = ch * 16u + 15u + j;                         // tile-relative END of that window
                    uint64_t fwd, rc;
                    sk_window_keys(rec, e, fwd, rc);
                    const uint64_t cn = fwd > rc ? fwd : rc;
                    uint32_t sl = sk_slot0(sk_khash(cn), table.mask);
                    for (;;) {
                        const sk_u4 en = table.slots[sl];
                        const uint64_t key = sk_slot_key(en);
                        if (key == cn) {
                            const uint32_t tp = en.w >> 1;
                            if (tp != 0x7FFFFFFFu) {
                                seed = true;
                                a_dir = (uint32_t)(fwd > rc) ^ (en.w & 1u);        // 0: the read runs along the strain, 1: against it
                                // text position of tile-relative stream offset x: along D + x, against D - x
                                const uint32_t xs = e - 30u;                       // the window's first base (wraps below 0: fine, mod 2^32)
                                a_diag = a_dir ? tp + 30u + xs : tp - xs;
                            }
                            break;
                        }
                        if (key == SK_EMPTY64) break;
                        sl = (sl + 1u) & table.mask;
                    }
                }
                if (pass == 1 || !SK_SEED2_MIN || TALLY || nq < (uint32_t)SK_SEED2_MIN) break;   // (the TALLY kernels are short of scalar registers: one try there)
                // A stretch none of whose seed windows is a k-mer of the strain (a differing base in it: 14 % of the windows at 0.5 %
                // substitutions) used to send all its windows down the one-by-one path.  One more try first, with the window furthest from
                // the first seed: the FIRST live window of the stretch's LAST chunk.  (wave-uniform: only when some stretch needs it)
                const unsigned long long sm1 = __ballot(seed);
                const unsigned long long span1 = (hi_lane == 63u ? ~0ull : ((2ull << hi_lane) - 1ull)) & ~((1ull << lo_lane) - 1ull);
                const bool need = act & (live != 0u) & ((sm1 & span1) == 0ull);
                if (!__ballot(need)) break;
                anchor = need & (lane == hi_lane);
                use_first = true;
            }
            // ---- every chunk takes the nearest seed of its stretch
            const unsigned long long sm = __ballot(seed);
            const unsigned long long span = (hi_lane == 63u ? ~0ull : ((2ull << hi_lane) - 1ull)) & ~((1ull << lo_lane) - 1ull);
            const unsigned long long cand = sm & span;
            const unsigned long long below = cand & ((2ull << lane) - 1ull);   // (itself included)
            const uint32_t src = below ? 63u - (uint32_t)__clzll(below) : cand ? (uint32_t)__builtin_ctzll(cand) : lane;
            const uint32_t dg = (uint32_t)__shfl((int)a_diag, (int)src);
            const uint32_t dir = (uint32_t)__shfl((int)a_dir, (int)src);
            bool has = act & (cand != 0ull) & (live != 0u);
            // ---- verify: the 48 bases of chunks ch-1, ch, ch+1 against the text on the diagonal
            uint32_t hits = 0u, bits16 = 0u, r0 = 0u;                          // (in ascending text position: bit a)
            const uint32_t x0 = ch * 16u - 16u;                                // tile-relative offset of chunk ch-1
            const uint32_t tq = dir ? dg - x0 - 47u : dg + x0;                 // lowest text position of the 48
            has = has && tq < table.text_bases && tq + 48u <= table.text_bases;
            if (has) {
                const uint32_t wi = tq >> 4, o2 = 2u * (tq & 15u);
                const uint32_t t0 = table.text2[wi], t1 = table.text2[wi + 1u], t2 = table.text2[wi + 2u], t3 = table.text2[wi + 3u];
                const uint32_t a0 = (uint32_t)(((((uint64_t)t0 << 32) | t1) << o2) >> 32);
                const uint32_t a1 = (uint32_t)(((((uint64_t)t1 << 32) | t2) << o2) >> 32);
                const uint32_t a2 = (uint32_t)(((((uint64_t)t2 << 32) | t3) << o2) >> 32);
                const uint32_t cid = ch + SK_SPAN_CH;                          // record 0 = the 8 chunks before the tile
                const uint32_t w0 = rec[((cid - 1u) >> 3) * SK_REC_DW + ((cid - 1u) & 7u)];
                const uint32_t w1 = rec[(cid >> 3) * SK_REC_DW + (cid & 7u)];
                const uint32_t w2 = rec[((cid + 1u) >> 3) * SK_REC_DW + ((cid + 1u) & 7u)];
                const uint32_t m0 = sk_mismatch16(w0 ^ (dir ? sk_revcomp32(a2) : a0)) | inv_prev;
                const uint32_t m1 = sk_mismatch16(w1 ^ (dir ? sk_revcomp32(a1) : a1));
                const uint32_t m2 = sk_mismatch16(w2 ^ (dir ? sk_revcomp32(a0) : a2)) | inv_next;
                const uint64_t good = ~((uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32)) & 0x0000FFFFFFFFFFFFull;
                uint64_t rr = good & (good << 1);
                rr &= rr << 2;
                rr &= rr << 4;
                rr &= rr << 8;
                rr &= rr << 15;                                                // runs of >= 31
                const uint32_t ver16 = (uint32_t)(rr >> 31) & live;            // bit j: the window ending at chunk start + 15 + j
                // rank map: the 16 windows start at 16 consecutive text positions from qmin up
                const uint32_t qmin = tq + 1u;
                const sk_u4 ra = table.rank[qmin >> 6], rb = table.rank[(qmin >> 6) + 1u];
                const uint32_t off = qmin & 63u;
                const uint64_t ma = ((uint64_t)ra.z << 32) | ra.y, mb = ((uint64_t)rb.z << 32) | rb.y;
                bits16 = (uint32_t)((ma >> off) | (off ? mb << (64u - off) : 0ull)) & 0xFFFFu;
                r0 = ra.x + (uint32_t)__popcll(ma & ((1ull << off) - 1ull));
                const uint32_t asc = dir ? __builtin_bitreverse32(ver16) >> 16 : ver16;
                hits = asc & bits16;
                const uint32_t dups = asc & ~bits16;                           // a k-mer of the strain, but its row is elsewhere
                uint32_t un = live & ~ver16;                                   // live windows the diagonal does not explain
                if (un && SK_CHUNK_REJECT) {
                    // Mostly a base that differs from the strain (a read error, a diverged genome): ~31 windows in a row hold
                    // it.  Before they go to the one-by-one path (two filter questions each), three questions for all of
                    // them: the 16-mers that start 15 and 8 bases before the differing base and at it.  Every window that
                    // holds the base holds one of the three, and a window that holds a 16-mer the strain does not have is
                    // no k-mer of the strain (the filter has no false negatives) -- wherever in the strain it might lie.
                    const uint64_t inval48 = (uint64_t)inv_prev | ((uint64_t)inv_next << 32);
                    uint64_t mmw = (((uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32)) & ~inval48) & 0x0000FFFFFFFFFFFFull;
                    uint32_t rej = 0;
                    for (int round = 0; round < 2 && mmw; round++) {          // (the first two differing bases; more: the one-by-one path)
                        const uint32_t x = (uint32_t)__builtin_ctzll(mmw);
                        mmw &= mmw - 1ull;
                        uint32_t p3[3], g3[3];
                        uint2 q3[3];
                        bool ok3[3];
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const int back = k == 0 ? 15 : k == 1 ? 8 : 0;
                            const uint32_t p = (int)x - back < 0 ? 0u : x - (uint32_t)back > 32u ? 32u : x - (uint32_t)back;   // 16-mer [p, p+16) of the 48 bases
                            p3[k] = p;
                            ok3[k] = ((inval48 >> p) & 0xFFFFull) == 0ull && p >= 1u;            // all ACGT, and inside at least one window
                            const uint32_t q = p >> 4, o2 = 2u * (p & 15u);
                            const uint32_t hiw = q == 0u ? w0 : q == 1u ? w1 : w2, low = q == 0u ? w1 : q == 1u ? w2 : 0u;
                            const uint32_t v16 = (uint32_t)(((((uint64_t)hiw << 32) | low) << o2) >> 32);
                            const uint32_t r16 = sk_revcomp32(v16);
                            g3[k] = sk_gmix(v16 < r16 ? v16 : r16);
                            q3[k] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                            if (ok3[k]) q3[k] = table.grid1[sk_grid1_block(g3[k], table.grid1_blocks)];
                        }
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            if (ok3[k] && !sk_grid_test(q3[k], sk_grid1_bits(g3[k]))) {
                                // windows j = p-16 .. p-1 (window j spans bases 1+j .. 31+j) hold the 16-mer [p, p+16)
                                const uint32_t lo = p3[k] > 16u ? p3[k] - 16u : 0u, hi = p3[k] - 1u > 15u ? 15u : p3[k] - 1u;
                                rej |= ((2u << hi) - 1u) & ~((1u << lo) - 1u);
                            }
                    }
                    un &= ~rej;
                }
                fb = un | (dir ? __builtin_bitreverse32(dups) >> 16 : dups);
            }
            if (TALLY && UNION) {
                // union table: the windows' rows each name their strains; consecutive rows mostly name the same ones, and so do
                // the chunks of one read (neighbouring lanes).  A chunk whose hits all name the same members (m1) hands its
                // count to the run of lanes with the same record and members -- one atomic per member and run --; a chunk
                // with several runs of members credits them itself.  Informative hits: one log entry (position, global row)
                // each, gathered in the workgroup's LDS share.
                const uint32_t nh = (uint32_t)__popc(hits);
                uint32_t rcd = 0xFFFFFF00u | lane, ihm = 0u, m1 = 0u, c1 = 0u;
                if (nh) {
                    rcd = sk_record_of(sink, (uint32_t)tile0 + ch * 16u);
                    uint32_t h = hits, cur = 0u, cnt = 0u;
                    bool several = false;
                    while (h) {
                        const uint32_t a = (uint32_t)__builtin_ctz(h);
                        h &= h - 1u;
                        const uint32_t row = r0 + (uint32_t)__popc(bits16 & ((1u << a) - 1u));
                        const uint2 mk = sk_umask(sink)[row];
                        if (mk.x != cur) {
                            if (cnt) { sk_union_credit(sink, rcd, cur, cnt); several = true; }
                            cur = mk.x; cnt = 0u;
                        }
                        cnt++;
                        if (mk.y) { ihm |= 1u << a; sk_union_informative(sink, rcd, mk.y); }
                    }
                    if (several) sk_union_credit(sink, rcd, cur, cnt);
                    else { m1 = cur; c1 = cnt; }
                }
                const uint32_t ni = (uint32_t)__popc(ihm);
                const uint32_t prev_r = (uint32_t)__shfl_up((int)rcd, 1), prev_m = (uint32_t)__shfl_up((int)m1, 1);
                const bool first = (lane == 0u) | (rcd != prev_r) | (m1 != prev_m);
                const unsigned long long fm = __ballot(first);
                const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
                const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
                const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
                const unsigned long long ltm = (1ull << lane) - 1ull;
                uint32_t sum_c = 0u, off_i = 0u, tot_i = 0u;
#pragma unroll
                for (int b = 0; b < 5; b++) {
                    const unsigned long long bc = __ballot((c1 >> b) & 1u), bi = __ballot((ni >> b) & 1u);
                    sum_c += (uint32_t)__popcll(bc & seg) << b;
                    off_i += (uint32_t)__popcll(bi & ltm) << b;
                    tot_i += (uint32_t)__popcll(bi) << b;
                }
                if (first && m1) sk_union_credit(sink, rcd, m1, sum_c);
                if (tot_i) {                                                             // (wave-uniform)
                    bool in_lds;
                    unsigned long long at = sk_log_reserve(sink, tot_i, lane, &in_lds) + off_i;
                    uint32_t h = ihm;
                    while (h) {
                        const uint32_t a = (uint32_t)__builtin_ctz(h);
                        h &= h - 1u;
                        const uint32_t row = r0 + (uint32_t)__popc(bits16 & ((1u << a) - 1u));
                        sk_log_put(sink, in_lds, at, make_uint2((uint32_t)tile0 + ch * 16u + 15u + (dir ? 15u - a : a), row));
                        at++;
                    }
                }
            } else if (TALLY) {
                // A chunk's windows all lie in the record that holds the chunk, and their rows are consecutive ranks:
                // the chunk adds popcount(hits) to its record's tally, and the informative ones among them come from 16
                // bits of the "row is informative" bitmap.  Lanes of one read are neighbours: one atomic per run.
                const uint32_t nh = (uint32_t)__popc(hits);
                uint32_t rcd = 0xFFFFFF00u | lane, ih = 0u;                // record (distinct per lane without a hit)
                if (nh) {
                    rcd = sk_record_of(sink, (uint32_t)tile0 + ch * 16u);
                    const uint32_t w = r0 >> 5, sh = r0 & 31u;
                    const uint64_t two = ((uint64_t)sink.infbits[w + 1u] << 32) | sink.infbits[w];
                    const uint32_t byrank = (uint32_t)(two >> sh) & 0xFFFFu;            // bit k: row r0 + k is informative
                    uint32_t bypos = byrank;                                             // bit a: the row of position a is
                    if (bits16 != 0xFFFFu) {
                        bypos = 0u;
                        for (uint32_t a = 0, k = 0; a < 16u; a++)
                            if ((bits16 >> a) & 1u) { bypos |= ((byrank >> k) & 1u) << a; k++; }
                    }
                    ih = hits & bypos;
                }
                const uint32_t ni = (uint32_t)__popc(ih);
                // sums over the run of lanes with the same record, and the offsets into the log: bit-sliced ballots
                const uint32_t prev = (uint32_t)__shfl_up((int)rcd, 1);
                const bool first = (lane == 0u) | (rcd != prev);
                const unsigned long long fm = __ballot(first);
                const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
                const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
                const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
                const unsigned long long ltm = (1ull << lane) - 1ull;
                uint32_t sum_h = 0u, sum_i = 0u, off_i = 0u, tot_i = 0u;
#pragma unroll
                for (int b = 0; b < 5; b++) {
                    const unsigned long long bh = __ballot((nh >> b) & 1u), bi = __ballot((ni >> b) & 1u);
                    sum_h += (uint32_t)__popcll(bh & seg) << b;
                    sum_i += (uint32_t)__popcll(bi & seg) << b;
                    off_i += (uint32_t)__popcll(bi & ltm) << b;
                    tot_i += (uint32_t)__popcll(bi) << b;
                }
                if (first && nh) {
                    atomicAdd(&sink.tally[2u * rcd], sum_h);
                    if (sum_i) atomicAdd(&sink.tally[2u * rcd + 1u], sum_i);
                }
                if (tot_i) {                                                             // (wave-uniform)
                    bool in_lds;
                    unsigned long long at = sk_log_reserve(sink, tot_i, lane, &in_lds) + off_i;
                    uint32_t h = ih;
                    while (h) {
                        const uint32_t a = (uint32_t)__builtin_ctz(h);
                        h &= h - 1u;
                        const uint32_t row = r0 + (uint32_t)__popc(bits16 & ((1u << a) - 1u));
                        sk_log_put(sink, in_lds, at, make_uint2((uint32_t)tile0 + ch * 16u + 15u + (dir ? 15u - a : a), sink.inv ? sink.inv[row] : row));
                        at++;
                    }
                }
            } else {
                // runs of verified windows -> the difference array; a run that goes on in the neighbouring chunk
                // (same diagonal) leaves out the two ends that would cancel
                const uint32_t lo_open = hits & 1u, hi_open = (hits >> 15) & 1u;
                const uint32_t pack = (has ? 1u : 0u) | (dir << 1) | (lo_open << 2) | (hi_open << 3);
                const uint32_t n_ch = (uint32_t)__shfl_down((int)ch, 1), n_dg = (uint32_t)__shfl_down((int)dg, 1), n_pk = (uint32_t)__shfl_down((int)pack, 1);
                const uint32_t p_ch = (uint32_t)__shfl_up((int)ch, 1), p_dg = (uint32_t)__shfl_up((int)dg, 1), p_pk = (uint32_t)__shfl_up((int)pack, 1);
                const bool cont_next = has && lane < 63u && n_ch == ch + 1u && n_dg == dg && (n_pk & 3u) == (1u | (dir << 1));
                const bool cont_prev = has && lane > 0u && p_ch + 1u == ch && p_dg == dg && (p_pk & 3u) == (1u | (dir << 1));
                // along the strain the next chunk lies at higher text positions, against it at lower ones
                const bool merge_hi = dir ? (cont_prev && hi_open && ((p_pk >> 2) & 1u)) : (cont_next && hi_open && ((n_pk >> 2) & 1u));
                const bool merge_lo = dir ? (cont_next && lo_open && ((n_pk >> 3) & 1u)) : (cont_prev && lo_open && ((p_pk >> 3) & 1u));
                uint32_t h = hits;
                while (h) {
                    const uint32_t s0 = (uint32_t)__builtin_ctz(h);
                    const uint32_t run = (uint32_t)__builtin_ctz(~(h >> s0));
                    const uint32_t i0 = r0 + (uint32_t)__popc(bits16 & ((1u << s0) - 1u));
                    if (!(s0 == 0u && merge_lo)) diff_add(i0, 1u);
                    if (!(s0 + run == 16u && merge_hi)) diff_add(i0 + run, 0xFFFFFFFFu);
                    h &= ~(((1u << run) - 1u) << s0);
                }
            }
        }
        // ---- the windows the diagonal did not settle: one by one (everything in this loop is wave-uniform)
        unsigned long long lanes = __ballot(fb != 0u);
        while (lanes) {
            const int l = __builtin_ctzll(lanes);
            lanes &= lanes - 1ull;
            const uint32_t f16 = (uint32_t)__builtin_amdgcn_readlane((int)fb, l);
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ch, l) * 16u + 15u;
            if (lane < 16u && ((f16 >> lane) & 1u))
                wq[qw + (uint32_t)__popc(f16 & ((1u << lane) - 1u))] = (uint16_t)(e0 + lane);
            qw += (uint32_t)__popc(f16);
            if (qw >= 128u) probe_windows(128u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (qw) probe_windows(qw);
    SK_PHASE(6);                                                     // stage 2
    if (bad) atomicAdd(&flags[0], 1u);
    if (!TALLY) {                                                 // the repeats of this tile, one atomic per index
        __syncthreads();
        for (uint32_t i = tid; i < SK_AGG; i += SK_THREADS) {
            const uint2 e = agg[i];
            if (e.y != 0u && ABLATE != 3) atomicAdd(&sink.diff[e.x], e.y);
        }
        SK_PHASE(7);                                                 // the closing barrier + flush
    } else {                                                      // the tile's share of the hit log: one atomic, one copy
        __syncthreads();
        const uint32_t nl = hl_n[0] < hl_n[1] ? hl_n[0] : hl_n[1];
        if (nl) {
            __shared__ unsigned long long gbase;
            if (tid == 0u) gbase = atomicAdd(sink.nhits, (unsigned long long)nl);
            __syncthreads();
            for (uint32_t i = tid; i < nl; i += SK_THREADS)
                if (gbase + i < sink.hits_cap) sink.hits[gbase + i] = agg[i];
        }
    }
}
