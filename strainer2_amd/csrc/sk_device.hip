// sk_device.hip -- device layer of libstrainer_kmer.so: hand-written HIP for CDNA4 / gfx950.
//
// Kernels (all integer/byte work; bound by L2 request rate and random-access rate, no MFMA):
//   sk_scan_grid      THE hot kernel (replaces reference src/genome_compare.c:213-229 +
//                     src/BIO_hash.c:161-172; in TALLY mode src/strain_detect.c:471-485):
//                       phase 1  bytes -> 2-bit code words + "not ACGT" masks in LDS (SWAR)
//                       phase 2  one filter lookup per aligned 16-base chunk (a 31-base window holds exactly one)
//                       stage 2  the windows of the surviving chunks: a few of them (anchors) are probed in the
//                                HBM table; a hit tells where the read lies on the strain, and the other windows
//                                are compared with the strain's 2-bit text there (seed and verify: 62-bit
//                                compares, as exact as a probe); atomicAdd on the row counter on a hit
//   sk_scan_wide      exact byte-string path for the rare windows that contain bytes other than
//                     ACGT (U / IUPAC / junk): the reference's signed-char orientation compare
//                     through its COMPLEMENT map.  Early exit unless phase 1 saw such a byte.
//   sk_table_insert   open-addressed key table (atomicCAS on the key word)
//   sk_grid_insert    the two filter levels (canonical 16-mers / 24-mers of every key)
//   sk_gather/scatter counters <-> caller row order
//
// Data layout in HBM:
//   slots  [S]  16 B  {u64 key, u32 counter index, u32 text position << 1 | orientation}: open addressing,
//                     linear probing, S = 2^s >= 2 nrows, empty = key all ones (one random line per hit)
//   grid1       8 B   blocks of the Bloom set of the strain's canonical 16-mers (level 1, sized for the L2)
//   grid2       8 B   blocks of the Bloom set of the strain's canonical 24-mers (level 2: the chunk + the 8 bases on either side)
//   text2       u32   the strain's bases, 2 bits each, records end to end (1.25 MB for 5 Mbp)
//   rank        16 B  per 64 text positions: {counter index of the first one, 64 "a row starts here" bits}
//   counts [ncols][nrows] u32, in locality (first-occurrence) order: one read's hits are adjacent
//   stream      u8    record stream: sequence bytes, records separated by '\n'
//
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include <unistd.h>
#include <pthread.h>
#include <errno.h>
#include <time.h>

#include "../../include/strainer_kmer.h"
#include "sk_rendezvous.h"
#include "sk_common.h"
#include "sk_internal.h"

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
// the number of strains: sk_union_resolve deals it out to the strains behind the scan)
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
template <bool TALLY, int ABLATE, bool CAND, bool UNION = false>
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
    if (!CAND) issue_loads(tile0);

    // ================= phase 1: bytes -> packed codes + invalid masks ==========================
    uint32_t candm = 0xFFu;                                        // this thread's chunks that are candidates (CAND)
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

// ---------------------------------------------------------------------------------------------
// The partitioned pipeline for large batches (the "LDS-staged probe"): the level-1 question "is this chunk's 16-mer in the
// strain at all?" costs one L2 request per chunk in sk_scan_grid, and the L2's request rate (about 270 G/s over its 128
// channels), not HBM, is what bounds that kernel.  Here the question is answered from LDS instead:
//   sk_bin        reads the stream once (the only full pass), decodes every chunk, hashes its canonical 16-mer and deals
//                 (chunk index in the tile, 20 hash bits) to one of SK_BIN_P partitions by 7 more hash bits: a fixed-size
//                 segment per (partition, tile) in HBM, 4 bytes per clean chunk = a quarter of the stream's bytes.
//   sk_lds_probe  one workgroup per partition (and share of the tiles) keeps that partition's slice of the filter
//                 -- a plain bitmap over the 20 bits, 128 KiB -- in LDS and streams the partition's segments past it;
//                 survivors (the strain's chunks plus a few per cent) set their byte in a per-chunk candidate map.
//   sk_scan_grid<.., CAND>  as before, but reading only the lines around candidates and asking the L2 filter only
//                 about them.
// Exactness is untouched: the filters only prune (no false negatives), stage 2 verifies what is left.
// ---------------------------------------------------------------------------------------------
#define SK_BIN_TILE   65536u
#define SK_BIN_CH     (SK_BIN_TILE / 16u)          // 4096 chunks: 12 bits
#define SK_BIN_P      128u                         // partitions: 7 bits
#define SK_BIN_CAP    40u                          // entries per (partition, tile): mean 28.5 clean chunks, +2 sigma; the rest go straight to the candidates
#define SK_BIN_WORDS  32768u                       // 2^20 bits per partition slice

// partition (7 bits) and in-partition key (20 bits) of a canonical 16-mer's mix; multiplier of its own, so that the slices'
// false positives are not the L2 filter's
__device__ __forceinline__ uint32_t sk_grid3_hash(uint32_t g) { return (g ^ (g >> 13)) * 0x5BD1E995u; }

__global__ void sk_grid3_insert(const sk_u4 *__restrict__ slots, uint64_t nslots, uint32_t *__restrict__ w3)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (from the resident table: built on first use only)
    if (i >= nslots) return;
    const uint64_t k = sk_slot_key(slots[i]);
    if (k == SK_EMPTY64) return;
    for (int off = 0; off < 16; off++) {
        const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
        const uint32_t r = sk_revcomp16(f);
        const uint32_t h = sk_grid3_hash(sk_gmix(f < r ? f : r));
        const uint32_t key = (h >> 5) & 0xFFFFFu;
        uint32_t *w = w3 + (size_t)(h >> 25) * SK_BIN_WORDS + (key >> 5);
        const uint32_t bit = 1u << (key & 31u);
        if (!(__builtin_nontemporal_load(w) & bit)) atomicOr(w, bit);
    }
}

__global__ __launch_bounds__(256)
void sk_bin(const uint8_t *__restrict__ stream, uint64_t nbytes, sk_table_view table, uint32_t *__restrict__ bins,
            uint8_t *__restrict__ bin_n, uint32_t ntiles, uint8_t *__restrict__ cand, uint32_t *__restrict__ flags)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[SK_BIN_P * SK_BIN_CAP];
    __shared__ uint32_t cnt[SK_BIN_P];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint64_t tile0 = (uint64_t)tile * SK_BIN_TILE;
    if (tid < SK_BIN_P) cnt[tid] = 0u;
    for (uint32_t i = tid; i < SK_BIN_P * SK_BIN_CAP; i += 256u) stage[i] = 0xFFFFFFFFu;     // "no entry"
    __syncthreads();
    uint32_t bad = 0;
    const bool inside = tile0 + SK_BIN_TILE <= nbytes;            // (workgroup-uniform)
#pragma unroll 1
    for (uint32_t half = 0; half < 2u; half++) {                  // 2 x 8 loads in flight per thread
        sk_u4 vv[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t c = tid + (half * 8u + (uint32_t)it) * 256u;
            vv[it] = inside ? sk_stream_load((const sk_u4 *)(stream + tile0 + (uint64_t)c * 16u))
                            : sk_load_chunk(stream, nbytes, (int64_t)(tile0 + (uint64_t)c * 16u));
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint32_t c = tid + (half * 8u + (uint32_t)it) * 256u;
            if (tile0 + (uint64_t)c * 16u >= nbytes) continue;
            const sk_u4 v = vv[it];
            uint32_t c0, c1, c2, c3, i0, i1, i2, i3;
            sk_decode4(v.x, c0, i0);
            sk_decode4(v.y, c1, i1);
            sk_decode4(v.z, c2, i2);
            sk_decode4(v.w, c3, i3);
            const uint32_t inv16 = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
            if (inv16) {
                if (sk_chunk_has_odd_byte(v, inv16)) {            // a byte for the byte-string kernel: note the chunk
                    bad = 1;
                    const uint32_t at = atomicAdd(&flags[2], 1u);
                    if (at < table.oddcap) table.oddlist[at] = (uint32_t)((tile0 >> 4) + c);
                }
                continue;
            }
            const uint32_t cw = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
            const uint32_t rc = sk_revcomp32(cw);
            const uint32_t h = sk_grid3_hash(sk_gmix(cw < rc ? cw : rc));
            const uint32_t p = h >> 25;
            const uint32_t ent = (c << 20) | ((h >> 5) & 0xFFFFFu);
            const uint32_t r = ent != 0xFFFFFFFFu ? atomicAdd(&cnt[p], 1u) : SK_BIN_CAP;
            if (r < SK_BIN_CAP) stage[p * SK_BIN_CAP + r] = ent;
            else cand[(tile0 >> 4) + c] = 1u;                     // no room in the segment: a candidate without being asked
        }
    }
    __syncthreads();
    // whole segments go out, 16 bytes per lane (SK_BIN_CAP is a multiple of 4): unused slots say "no entry"
    for (uint32_t i = tid; i < SK_BIN_P * SK_BIN_CAP / 4u; i += 256u) {
        const uint32_t p = i / (SK_BIN_CAP / 4u), r4 = i - p * (SK_BIN_CAP / 4u);
        ((sk_u4 *)(bins + ((size_t)p * ntiles + tile) * SK_BIN_CAP))[r4] = ((const sk_u4 *)stage)[i];
    }
    (void)bin_n;
    if (bad) atomicAdd(&flags[0], 1u);
}

__global__ __launch_bounds__(1024)
void sk_lds_probe(const uint32_t *__restrict__ w3, const uint32_t *__restrict__ bins, const uint8_t *__restrict__ bin_n,
                  uint32_t ntiles, uint32_t splits, uint8_t *__restrict__ cand)
{
    extern __shared__ uint32_t slice[];                           // SK_BIN_WORDS words = 128 KiB
    const uint32_t p = blockIdx.x / splits, sp = blockIdx.x % splits, tid = threadIdx.x;
    const sk_u4 *src = (const sk_u4 *)(w3 + (size_t)p * SK_BIN_WORDS);
    for (uint32_t i = tid; i < SK_BIN_WORDS / 4u; i += 1024u) ((sk_u4 *)slice)[i] = src[i];
    __syncthreads();
    const uint32_t t_lo = (uint32_t)((uint64_t)ntiles * sp / splits), t_hi = (uint32_t)((uint64_t)ntiles * (sp + 1u) / splits);
    const sk_u4 *seg = (const sk_u4 *)(bins + ((size_t)p * ntiles + t_lo) * SK_BIN_CAP);
    (void)bin_n;
    const uint32_t total4 = (t_hi - t_lo) * (SK_BIN_CAP / 4u);    // 16-byte groups of four entries
    uint8_t *const cbase = cand + (size_t)t_lo * SK_BIN_CH;
    auto judge = [&](uint32_t ent, uint32_t t) {
        const uint32_t key = ent & 0xFFFFFu;
        if (ent != 0xFFFFFFFFu && ((slice[key >> 5] >> (key & 31u)) & 1u)) cbase[(size_t)t * SK_BIN_CH + (ent >> 20)] = 1u;
    };
    for (uint32_t i0 = 0; i0 < total4; i0 += 4096u) {
        sk_u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + tid + 1024u * (uint32_t)k;
            v[k] = (sk_u4){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (i < total4) v[k] = __builtin_nontemporal_load(seg + i);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = (i0 + tid + 1024u * (uint32_t)k) / (SK_BIN_CAP / 4u);
            judge(v[k].x, t); judge(v[k].y, t); judge(v[k].z, t); judge(v[k].w, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wide (byte-string) path
// ---------------------------------------------------------------------------------------------
__constant__ signed char sk_comp_dev[256];

struct sk_wide_view {
    const char     *keys31;     // [nwide][32]
    const uint32_t *rows;       // [nwide]
    const uint32_t *index;      // [wmask+1]  0 = empty, else key index + 1
    uint32_t        wmask;
    uint32_t        nwide;
};

template <bool TALLY, bool UNION = false>
__global__ __launch_bounds__(256)
void sk_scan_wide(const uint8_t *__restrict__ stream, uint64_t nbytes, uint64_t emit_begin,
                  sk_table_view table, sk_wide_view wide, sk_sink sink,
                  const uint32_t *__restrict__ flags, uint32_t *__restrict__ next_flags)
{
    // the NEXT launch's flag words (the context alternates between two sets) are zeroed here, behind this launch's scan kernel and
    // before the next one's: a memset per scan less on the stream
    if (blockIdx.x == 0u && threadIdx.x < 4u) next_flags[threadIdx.x] = 0u;
    if (flags[0] == 0u) return;                        // no window with a non-ACGT byte in this batch
    // Work list: phase 1 of the scan kernel noted every 16-byte chunk that holds such a byte (flags[2] of them).
    // A window that needs this kernel contains one; it is handled from the chunk that holds its LAST non-ACGT
    // byte, so every window is handled once and the cost follows the number of odd bytes, not the batch size.
    // If the list overflowed, every position of the batch is visited instead.
    const uint32_t nodd = flags[2];
    const bool listed = nodd <= table.oddcap;
    const uint64_t nitems = listed ? (uint64_t)nodd * 46u : nbytes;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; item < nitems; item += stride) {
        uint64_t p, owner = ~0ull;
        if (listed) { owner = table.oddlist[item / 46u]; p = owner * 16u + item % 46u; }
        else p = item;
        if (p < emit_begin || p >= nbytes || p < (uint64_t)(SK_K - 1)) continue;
        const uint8_t *w = stream + (p - (SK_K - 1));
        char u[SK_K];
        bool hard = false, pure = true;
        int last_odd = -1;
        for (int i = 0; i < SK_K; i++) {
            const uint32_t b = w[i];
            hard |= (bool)sk_is_hard_break(b);
            if (!sk_is_acgt(b)) { pure = false; last_odd = i; }
            u[i] = (char)sk_upper(b);
        }
        if (hard || pure) continue;                    // skipped by the reference / done by the scan kernel
        if (listed && ((p - (SK_K - 1) + (uint64_t)last_odd) >> 4) != owner) continue;   // another chunk's window
        // orientation: sign of (window - revcomp) in signed-char order (src/genome_compare.c:1122-1141)
        int sign = 0;
        for (int i = 0; i < SK_K && sign == 0; i++) {
            const signed char f = (signed char)u[i];
            const signed char r = sk_comp_dev[(uint8_t)u[SK_K - 1 - i]];
            sign = (f > r) - (r > f);
        }
        char o[SK_K + 1];
        if (sign >= 0) { for (int i = 0; i < SK_K; i++) o[i] = u[i]; }
        else           { for (int i = 0; i < SK_K; i++) o[SK_K - 1 - i] = (char)sk_comp_dev[(uint8_t)u[i]]; }
        o[SK_K] = 0;
        bool opure = true, has_n = false, has_nul = false;
        for (int i = 0; i < SK_K; i++) {
            opure &= (bool)sk_is_acgt((uint8_t)o[i]) & ((uint8_t)o[i] < 'a');
            has_n |= (o[i] == 'N');
            has_nul |= (o[i] == 0);
        }
        if (has_n || has_nul) continue;
        if (opure) {                                   // e.g. a window with U whose revcomp wins
            uint64_t key = 0;
            for (int i = 0; i < SK_K; i++) key = (key << 2) | sk_code((uint8_t)o[i]);
            sk_probe<TALLY, false, UNION>(key, table, sink, (uint32_t)p);
        } else if (wide.nwide) {
            uint32_t slot = sk_hash_wide(o) & wide.wmask;
            for (;;) {
                const uint32_t e = wide.index[slot];
                if (e == 0u) break;
                const char *cand = wide.keys31 + (size_t)(e - 1u) * 32u;
                bool same = true;
                for (int i = 0; i < SK_K; i++) same &= (cand[i] == o[i]);
                if (same) { sk_on_hit<TALLY, false, UNION>(sink, wide.rows[e - 1u], (uint32_t)p); break; }
                slot = (slot + 1u) & wide.wmask;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// table build
// ---------------------------------------------------------------------------------------------
__global__ void sk_fill64(uint64_t *p, uint64_t n, uint64_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void sk_table_insert(const uint64_t *__restrict__ in, uint32_t n, sk_u4 *slots, uint32_t mask, uint32_t *flags,
                                const uint32_t *__restrict__ perm, const uint32_t *__restrict__ locality)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    const uint32_t idx = perm ? perm[i] : i;
    if (k == SK_EMPTY64) return;                       // wide row: not in this table
    if (k > SK_KMASK62) { atomicAdd(&flags[1], 1u); return; }
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot],
                                                (unsigned long long)SK_EMPTY64, (unsigned long long)k);
        if (old == SK_EMPTY64) {
            ((uint32_t *)&slots[slot])[2] = idx;
            ((uint32_t *)&slots[slot])[3] = 0xFFFFFFFEu | (locality ? locality[i] >> 31 : 0u);   // no text position (yet)
            return;
        }
        if (old == k) { atomicAdd(&flags[1], 1u); return; }     // duplicate key
        slot = (slot + 1u) & mask;
    }
}

// TALLY: the records with at least one hit, as {record, all hits, informative hits} (unordered); *n counts them
__global__ void sk_tally_compact(const uint32_t *__restrict__ tally, uint32_t nrec, uint32_t *__restrict__ out, unsigned long long *n)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint2 t = r < nrec ? ((const uint2 *)tally)[r] : make_uint2(0u, 0u);
    const unsigned long long m = __ballot(t.x != 0u);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(m);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(n, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), (int)leader) << 32) | (uint32_t)__shfl((int)(uint32_t)base, (int)leader);
    if (t.x != 0u) {
        uint32_t *o = out + 3ull * (base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull)));
        o[0] = r; o[1] = t.x; o[2] = t.y;
    }
}

// TALLY: one bit per counter index, set where the type column holds `value`
__global__ void sk_inf_bitmap(const uint32_t *__restrict__ type, uint32_t n, uint32_t value, uint32_t *__restrict__ bits)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool is = i < n && type[i] == value;
    const unsigned long long m = __ballot(is);
    if ((threadIdx.x & 63u) == 0u && i < n + 64u) { bits[2u * (i >> 6)] = (uint32_t)m; bits[2u * (i >> 6) + 1u] = (uint32_t)(m >> 32); }
}

// text position of every row into its table slot (pos_by_idx: by counter index, 0xFFFFFFFF = none)
__global__ void sk_table_setpos(sk_u4 *slots, uint64_t nslots, const uint32_t *__restrict__ pos_by_idx)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = slots[i];
        if (sk_slot_key(e) == SK_EMPTY64) continue;
        ((uint32_t *)&slots[i])[3] = (pos_by_idx[e.z] << 1) | (e.w & 1u);
    }
}

// ---- the difference array of the column being scanned, folded into it: counts[i] += diff[0] + ... + diff[i] ----
#define SK_DIFF_PER_BLOCK 4096u                        // 256 threads x 16 entries
__global__ __launch_bounds__(256) void sk_diff_block_sums(const uint32_t *__restrict__ diff, uint32_t n, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * SK_DIFF_PER_BLOCK + threadIdx.x * 16u;
    uint32_t t = 0;
    for (uint32_t i = 0; i < 16u; i++) if (base + i < n) t += diff[base + i];
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 128u; d > 0u; d >>= 1) { if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(1024) void sk_diff_scan_sums(uint32_t *sums, uint32_t nb)     // one block: exclusive scan in place
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023u) / 1024u, lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t i = lo; i < lo + per && i < nb; i++) t += sums[i];
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - t;              // exclusive prefix of this thread's slice
    for (uint32_t i = lo; i < lo + per && i < nb; i++) { const uint32_t v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void sk_diff_apply(uint32_t *__restrict__ diff, uint32_t n, const uint32_t *__restrict__ sums,
                                                      uint32_t *__restrict__ counts, uint32_t nrows)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * SK_DIFF_PER_BLOCK + threadIdx.x * 16u;
    uint32_t v[16], t = 0;
    for (uint32_t i = 0; i < 16u; i++) { v[i] = base + i < n ? diff[base + i] : 0u; t += v[i]; }
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 256u; d <<= 1) {
        const uint32_t u = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += u;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - t;
    for (uint32_t i = 0; i < 16u; i++) {
        run += v[i];
        if (base + i < nrows && run) counts[base + i] += run;
        if (base + i < n && v[i]) diff[base + i] = 0u;
    }
}

// counter columns live in "locality order" on the device (perm: caller's row -> counter index)
__global__ void sk_gather_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void sk_scatter_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}

// the permutation is the caller's locality[] without its orientation bit
__global__ void sk_perm_from_locality(uint32_t *__restrict__ perm, const uint32_t *__restrict__ locality, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = locality[i] & 0x7FFFFFFFu;
}

__global__ void sk_invert_perm(uint32_t *__restrict__ inv, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (perm[i] >= n) { atomicAdd(&flags[1], 1u); return; }
    inv[perm[i]] = i;
}

// grid filters: the canonical form of every 16-mer of every key, into both levels
__global__ void sk_grid_insert(const uint64_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ w1, uint32_t nblocks1,
                               uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    if (k == SK_EMPTY64) return;
    for (int off = 0; off < 16; off++) {
        const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
        const uint32_t r = sk_revcomp16(f);
        const uint32_t g = sk_gmix(f < r ? f : r);
        const uint32_t a = sk_grid1_bits(g);
        // consecutive keys share 15 of their 16 sub-words: most bits are set already, so look before the atomic
        uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
        const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
        if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
        if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
    }
    for (int off = 0; off < 8; off++) sk_grid2_insert24(w2, shift2, (k >> (2 * (7 - off))) & 0xFFFFFFFFFFFFull);   // level 2: its eight 24-mers
}

// ---- the table built ON THE DEVICE from the strain's 2-bit text (sk_table_build_from_text; strain_detect's opening) -----------------
// The host hands over the text (records end to end) and one bit per position "a window of 31 A/C/G/T bases of one record starts
// here" (src/genome_compare.c:1000-1019: every such window's oriented form is a key).  Rows are numbered by first occurrence along
// the text, which makes the counter index of a text position its RANK -- the layout the scan's verify stage wants anyway.
__device__ __forceinline__ bool sk_bit(const uint32_t *__restrict__ bits, uint32_t p) { return (bits[p >> 5] >> (p & 31u)) & 1u; }
__device__ __forceinline__ uint64_t sk_text_canon(const uint32_t *__restrict__ text2, uint32_t p, uint32_t *is_fwd)
{
    const uint64_t fwd = sk_text_key(text2, p);
    uint64_t r = ((uint64_t)__builtin_bitreverse32((uint32_t)fwd) << 32) | __builtin_bitreverse32((uint32_t)(fwd >> 32));
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    const uint64_t rc = (~r) >> 2;
    *is_fwd = fwd > rc;
    return fwd > rc ? fwd : rc;
}
// every window's key into the slots; the slot keeps the LOWEST position of its key (with the orientation it has there)
__global__ void sk_build_insert(const uint32_t *__restrict__ text2, const uint32_t *__restrict__ startok, uint32_t nbases, sk_u4 *slots, uint32_t mask)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases || !sk_bit(startok, p)) return;
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot], (unsigned long long)SK_EMPTY64, (unsigned long long)k);
        if (old == SK_EMPTY64 || old == k) { atomicMin(&((uint32_t *)&slots[slot])[3], (p << 1) | fw); return; }
        slot = (slot + 1u) & mask;
    }
}
__device__ __forceinline__ uint32_t sk_build_find(const sk_u4 *slots, uint32_t mask, uint64_t k)
{
    uint32_t slot = sk_slot0(sk_khash(k), mask);
    while (sk_slot_key(slots[slot]) != k) slot = (slot + 1u) & mask;       // (the key is there: sk_build_insert put it)
    return slot;
}
// the positions at which a key occurs for the first time: the rank map's bits
__global__ void sk_build_first(const uint32_t *__restrict__ text2, const uint32_t *__restrict__ startok, uint32_t nbases, const sk_u4 *slots, uint32_t mask, sk_u4 *rank)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases || !sk_bit(startok, p)) return;
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    if ((slots[sk_build_find(slots, mask, k)].w >> 1) != p) return;
    atomicOr((uint32_t *)&rank[p >> 6] + ((p & 32u) ? 2 : 1), 1u << (p & 31u));          // (.y: bits 0..31 of the block, .z: 32..63)
}
// rank[b].x = first-occurrence positions before block b (one workgroup walks the blocks in slices; *total = all of them)
__global__ __launch_bounds__(1024) void sk_build_rank_scan(sk_u4 *rank, uint32_t nblk, uint32_t *total)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nblk + 1023u) / 1024u, lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) t += (uint32_t)__popc(rank[i].y) + (uint32_t)__popc(rank[i].z);
    part[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - t;
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const uint32_t v = (uint32_t)__popc(rank[i].y) + (uint32_t)__popc(rank[i].z); rank[i].x = run; run += v; }
    if (threadIdx.x == 1023u) *total = part[1023];
}
// every row's counter index (= the rank of its first position) into its slot, and its key into the row-ordered key list
__global__ void sk_build_index(const uint32_t *__restrict__ text2, uint32_t nbases, sk_u4 *slots, uint32_t mask, const sk_u4 *__restrict__ rank, uint64_t *__restrict__ keys_by_row)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p + SK_K > nbases) return;
    const sk_u4 r = rank[p >> 6];
    const uint64_t bits = ((uint64_t)r.z << 32) | r.y;
    if (!((bits >> (p & 63u)) & 1ull)) return;
    const uint32_t idx = r.x + (uint32_t)__popcll(bits & ((1ull << (p & 63u)) - 1ull));
    uint32_t fw;
    const uint64_t k = sk_text_canon(text2, p, &fw);
    ((uint32_t *)&slots[sk_build_find(slots, mask, k)])[2] = idx;
    keys_by_row[idx] = k;
}
__global__ void sk_set_rows_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ rows, uint32_t n, const uint32_t *__restrict__ perm, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm ? perm[rows[i]] : rows[i]] = v;
}
__global__ void sk_gather_keys(uint64_t *__restrict__ out, const uint64_t *__restrict__ keys_by_row, const uint32_t *__restrict__ rows, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keys_by_row[rows[i]];
}
__global__ void sk_fill32(uint32_t *p, uint32_t n, uint32_t v)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the same set from the table's slots (a table loaded without a text stage: built when the first scan needs it)
__global__ void sk_grid_insert_slots(const sk_u4 *__restrict__ slots, uint64_t nslots, uint32_t *__restrict__ w1, uint32_t nblocks1,
                                     uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const uint64_t k = sk_slot_key(slots[i]);
        if (k == SK_EMPTY64) continue;
        for (int off = 0; off < 16; off++) {
            const uint32_t f = (uint32_t)(k >> (2 * (15 - off)));
            const uint32_t r = sk_revcomp16(f);
            const uint32_t g = sk_gmix(f < r ? f : r);
            const uint32_t a = sk_grid1_bits(g);
            uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
            const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
            if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
            if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
        }
        for (int off = 0; off < 8; off++) sk_grid2_insert24(w2, shift2, (k >> (2 * (7 - off))) & 0xFFFFFFFFFFFFull);
    }
}

// ... and from the strain's TEXT, when every key is a window of it: a key's sixteen 16-mers are the text's 16-mers at its
// place and the fifteen places behind it, and consecutive keys share fifteen of them -- inserting every 16-mer of the text ONCE
// does in nbases steps what the key-wise kernels do in 16 x nrows (5.5 ms -> 0.3 ms for a 5 Mbp strain; 167 ms -> 10 ms for the
// union of 32).  16-mers that no key holds (across an N, across two records) only add a few false positives.
__global__ void sk_grid_insert_text(const uint32_t *__restrict__ text2, uint32_t nbases, uint32_t *__restrict__ w1, uint32_t nblocks1,
                                    uint32_t *__restrict__ w2, uint32_t shift2)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q + 16u > nbases) return;
    const uint32_t w = q >> 4, o2 = 2u * (q & 15u);
    const uint32_t f = (uint32_t)(((((uint64_t)text2[w] << 32) | text2[w + 1u]) << o2) >> 32);
    const uint32_t r = sk_revcomp16(f);
    const uint32_t g = sk_gmix(f < r ? f : r);
    const uint32_t a = sk_grid1_bits(g);
    uint32_t *blk = w1 + 2u * (size_t)sk_grid1_block(g, nblocks1);
    const uint32_t m0 = (1u << ((a >> 24) & 31u)) | (1u << ((a >> 16) & 31u)), m1 = (1u << ((a >> 8) & 31u)) | (1u << (a & 31u));
    if ((__builtin_nontemporal_load(&blk[0]) & m0) != m0) atomicOr(&blk[0], m0);
    if ((__builtin_nontemporal_load(&blk[1]) & m1) != m1) atomicOr(&blk[1], m1);
    if (q + 24u <= nbases) sk_grid2_insert24(w2, shift2, sk_text_24(text2, q));       // level 2: the text's 24-mer at this place
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
#define SK_STAGE_BYTES   (64ull << 20)
#define SK_NSTAGE        2

struct sk_pin { void *p; size_t n; bool used; bool registered; };    // registered: malloc'd + hipHostRegister; else hipHostMalloc

struct sk_ctx {
    int          device;
    hipStream_t  stream;
    std::vector<sk_pin> pins;         // page-locked host buffers of sk_pinned_alloc, kept for re-use
    pthread_mutex_t pin_mu;
    // table
    sk_u4       *d_keys;              // the slot array (name kept: "is a table loaded" checks)
    uint32_t     slots_log2;
    uint32_t    *d_text2;                 // seed and verify: the strain's text and the rank map (sk_table_load_text)
    sk_u4       *d_rank;
    uint32_t     text_bases;
    uint2       *d_grid1, *d_grid2;       // grid kernel's two filter levels
    uint32_t     grid1_blocks, grid2_blocks_log2;
    long         grid_kib;                // option: size of level 1 in KiB (-1 = automatic)
    uint64_t    *d_keys_by_row;           // sk_table_build_from_text: the keys in row order, until sk_table_export_keys fetched them
    bool         grid_pending;            // the two levels are allocated and zeroed, not filled yet (sk_grid_ensure)
    bool         all_rows_in_text;        // every key is a window of the text stage (sk_table_load_text)
    long         odd_cap;                 // option (tests): usable length of the odd-chunk list, 0 = all of it
    uint32_t     nrows, ncols;
    uint32_t    *d_counts;
    uint32_t    *d_perm, *d_inv;      // locality order of the counters (NULL = caller's row order)
    uint32_t    *d_locality;          // the caller's locality[] as given (with the orientation bit)
    uint32_t    *d_tmp;               // [nrows] scratch for fetch/set through the permutation
    uint32_t    *d_infbits;           // TALLY: informative-row bitmap of (infbits_col, infbits_val); valid while infbits_ok
    uint32_t     infbits_col, infbits_val; bool infbits_ok;
    uint32_t    *d_diff, *d_diff_sums;// difference array of the column being scanned [nrows + 1] and its block sums
    int          diff_col;            // the column d_diff belongs to, -1 = nothing pending
    std::vector<uint32_t> h_perm;     // host copy of the permutation (empty = identity)
    // wide keys
    char        *d_wide_keys;
    uint32_t    *d_wide_rows;
    uint32_t    *d_wide_index;
    uint32_t     wide_mask, nwide;
    // staging for host-resident streams
    uint8_t     *h_stage[SK_NSTAGE];
    uint8_t     *d_stage[SK_NSTAGE];
    hipEvent_t   stage_done[SK_NSTAGE];
    int          stage_next;
    hipEvent_t   copied[64];           // ring of "host buffer of ticket t has been read" events
    uint64_t     tickets;              // tickets issued so far
    // flags: [0] wide windows seen in the current batch, [1] table build errors
    uint32_t    *d_flags;
    uint32_t     flag_set;            // which of the two sets of scan flags (words 0..3 / 4..7) the launch in hand uses
    bool         flags_ready;         // both sets are zero (false after anything else wrote to the flag block)
    uint32_t    *d_oddlist;           // chunks with a byte for the byte-string kernel (SK_ODDCAP entries)
    // timing
    std::vector<hipEvent_t> ev;        // begin/end pairs of launches not yet added up: a ring of at most SK_EV_PAIRS
    std::vector<hipEvent_t> ev_free;   // pairs that have been added up, for the next launches
    double       timed_ms;
    uint64_t     timed_launches;
    // options
    long         table_load_pct;
    long         ablate;              // timing experiments: kernel variants that skip memory stages
    long         dev_uncached;        // experiment: sk_dev_alloc hands out memory the L2 does not keep
    uint32_t    *d_grid3;             // partitioned pipeline: the SK_BIN_P filter slices (bitmaps) of sk_lds_probe
    void        *p_bins, *p_binn, *p_cand;    // its grow-only scratch: segments, segment fills, candidate bytes
    size_t       p_bins_cap, p_binn_cap, p_cand_cap;
    long         pipeline;            // option: 0/1 = the single kernel (default), 2 = the partitioned pipeline (experiment)
    long         no_text;             // option "text_stage"=0: stage 2 probes every window on its own (A/B, tests)
    void        *t_tally, *t_hits, *t_compact;    // grow-only device scratch of the tally path
    size_t       t_tally_cap, t_hits_cap, t_compact_cap;
    uint8_t     *h_tally;             // pinned landing area of the tallies (+ the hit counter)
    size_t       h_tally_cap;
    uint32_t     t_inflight_nrec;     // a sk_tally_launch waiting for its sk_tally_collect
    uint64_t     t_inflight_cap;
    struct sk_batch *own_batch;       // sk_tally_batch's private batch
    void        *comm;                // RCCL communicator from sk_comm_init (NULL: single process)
    int          comm_rank, comm_world;
    char         err[512];
};

static int sk_fail(sk_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define SK_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return sk_fail((ctx), SK_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

extern "C" const char *sk_strerror(int code)
{
    switch (code) {
    case SK_OK: return "ok";
    case SK_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case SK_E_HIP: return "HIP runtime error";
    case SK_E_ARG: return "bad argument";
    case SK_E_NOMEM: return "out of memory";
    case SK_E_OPEN: return "could not open file";
    case SK_E_DUPKEY: return "duplicate or malformed key in table load";
    case SK_E_STATE: return "call out of order";
    case SK_E_RCCL: return "RCCL error";
    case SK_E_SPLIT: return "a file could not be cut at record boundaries";
    case SK_E_PLAN: return "the ranks computed different work plans";
    default: return "unknown error";
    }
}

extern "C" const char *sk_last_error(const sk_ctx *ctx) { return ctx ? ctx->err : "no context"; }

extern "C" int sk_ctx_create(sk_ctx **out, int device)
{
    if (!out) return SK_E_ARG;
    *out = NULL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SK_E_NODEVICE;
    if (device < 0 || device >= ndev) return SK_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return SK_E_NODEVICE;
    sk_ctx *c = new (std::nothrow) sk_ctx();
    if (!c) return SK_E_NOMEM;
    c->device = device;
    pthread_mutex_init(&c->pin_mu, NULL);
    c->table_load_pct = 50;
    c->grid_kib = -1;
    c->diff_col = -1;
    c->err[0] = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SK_E_NODEVICE; }
    if (hipMalloc((void **)&c->d_flags, 32 * sizeof(uint32_t)) != hipSuccess) { delete c; return SK_E_NOMEM; }   // (words 16..31: scratch of the small collectives)
    if (hipMalloc((void **)&c->d_oddlist, (size_t)SK_ODDCAP * sizeof(uint32_t)) != hipSuccess) { hipFree(c->d_flags); delete c; return SK_E_NOMEM; }
    hipMemsetAsync(c->d_flags, 0, 32 * sizeof(uint32_t), c->stream);
    {   // the complement map goes to the device once per process and device, not once per context: the copy to a symbol
        // waits for the device, and 32 strains opened at once (strain_detect -S) spent 0.19 s each in here
        static pthread_mutex_t once_mu = PTHREAD_MUTEX_INITIALIZER;
        static unsigned long long done_mask[4];
        pthread_mutex_lock(&once_mu);
        const bool have = device < 256 && ((done_mask[device >> 6] >> (device & 63)) & 1ull);
        hipError_t e = hipSuccess;
        if (!have) {
            signed char comp[256];
            sk_fill_complement(comp);
            e = hipMemcpyToSymbol(HIP_SYMBOL(sk_comp_dev), comp, sizeof comp);
            if (e == hipSuccess && device < 256) done_mask[device >> 6] |= 1ull << (device & 63);
        }
        pthread_mutex_unlock(&once_mu);
        if (e != hipSuccess) { delete c; return SK_E_NODEVICE; }
    }
    *out = c;
    return SK_OK;
}

static void sk_table_release(sk_ctx *c)
{
    hipFree(c->d_keys); c->d_keys = NULL;
    hipFree(c->d_text2); c->d_text2 = NULL;
    hipFree(c->d_rank); c->d_rank = NULL;
    c->text_bases = 0;
    hipFree(c->d_grid1); c->d_grid1 = NULL;
    hipFree(c->d_grid2); c->d_grid2 = NULL;
    hipFree(c->d_grid3); c->d_grid3 = NULL;
    hipFree(c->d_keys_by_row); c->d_keys_by_row = NULL;
    c->grid_pending = false; c->all_rows_in_text = false;
    hipFree(c->d_counts); c->d_counts = NULL;
    hipFree(c->d_perm); c->d_perm = NULL;
    c->h_perm.clear();
    hipFree(c->d_inv); c->d_inv = NULL;
    hipFree(c->d_locality); c->d_locality = NULL;
    hipFree(c->d_tmp); c->d_tmp = NULL;
    hipFree(c->d_diff); c->d_diff = NULL;
    hipFree(c->d_diff_sums); c->d_diff_sums = NULL;
    c->diff_col = -1;
    hipFree(c->d_infbits); c->d_infbits = NULL; c->infbits_ok = false;
    hipFree(c->d_wide_keys); c->d_wide_keys = NULL;
    hipFree(c->d_wide_rows); c->d_wide_rows = NULL;
    hipFree(c->d_wide_index); c->d_wide_index = NULL;
    c->nrows = c->ncols = c->nwide = 0;
}

extern "C" void sk_comm_destroy(sk_ctx *c);
extern "C" void sk_batch_destroy(struct sk_batch *b);

extern "C" void sk_ctx_destroy(sk_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    sk_comm_destroy(c);
    sk_table_release(c);
    for (int i = 0; i < SK_NSTAGE; i++) {
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->d_stage[i]) hipFree(c->d_stage[i]);
        if (c->stage_done[i]) hipEventDestroy(c->stage_done[i]);
    }
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_free) hipEventDestroy(e);
    for (int i = 0; i < 64; i++) if (c->copied[i]) hipEventDestroy(c->copied[i]);
    if (c->own_batch) sk_batch_destroy(c->own_batch);
    hipFree(c->t_tally); hipFree(c->t_hits); hipFree(c->t_compact);
    hipFree(c->p_bins); hipFree(c->p_binn); hipFree(c->p_cand);
    if (c->h_tally) hipHostFree(c->h_tally);
    for (sk_pin &q : c->pins) { if (q.registered) { hipHostUnregister(q.p); free(q.p); } else hipHostFree(q.p); }
    pthread_mutex_destroy(&c->pin_mu);
    hipFree(c->d_flags);
    hipFree(c->d_oddlist);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int sk_set_option(sk_ctx *c, const char *name, long value)
{
    if (!c || !name) return SK_E_ARG;
    if (!strcmp(name, "table_load_pct")) { if (value < 5 || value > 90) return SK_E_ARG; c->table_load_pct = value; return SK_OK; }
    if (!strcmp(name, "grid_kib")) { if (value < -1 || value == 0 || value > (1 << 22)) return SK_E_ARG; c->grid_kib = value; return SK_OK; }
    if (!strcmp(name, "odd_list_cap")) { if (value < 0 || value > (long)SK_ODDCAP) return SK_E_ARG; c->odd_cap = value; return SK_OK; }
    if (!strcmp(name, "dev_alloc_uncached")) { c->dev_uncached = value != 0; return SK_OK; }
    if (!strcmp(name, "text_stage")) { c->no_text = value == 0; return SK_OK; }
    if (!strcmp(name, "pipeline")) { if (value < 0 || value > 2) return SK_E_ARG; c->pipeline = value; return SK_OK; }
#ifdef SK_EXPERIMENTS
    if (!strcmp(name, "ablate")) { c->ablate = value; return SK_OK; }
#else
    // the timing variants of the hot kernel that leave memory stages out (some of them count wrongly) are compiled
    // only into an experiments build (make EXPERIMENTS=1): the product library has no switch that changes a count
    if (!strcmp(name, "ablate")) return value == 0 ? SK_OK : sk_fail(c, SK_E_ARG, "option \"ablate\" needs a library built with make EXPERIMENTS=1");
#endif
    return sk_fail(c, SK_E_ARG, "unknown option %s", name);
}

extern "C" int sk_table_load(sk_ctx *c, const uint64_t *keys, uint32_t nrows, uint32_t ncols)
{
    return sk_table_load_ex(c, keys, nrows, ncols, NULL);
}

extern "C" int sk_table_load_ex(sk_ctx *c, const uint64_t *keys, uint32_t nrows, uint32_t ncols, const uint32_t *locality)
{
    if (!c || (!keys && nrows) || ncols == 0 || ncols > 16) return SK_E_ARG;
    if (locality) {                                    // must be a permutation of 0..nrows-1
        std::vector<uint8_t> seen(nrows, 0);
        for (uint32_t i = 0; i < nrows; i++) {
            const uint32_t l = locality[i] & 0x7FFFFFFFu;
            if (l >= nrows || seen[l]) return sk_fail(c, SK_E_ARG, "locality is not a permutation");
            seen[l] = 1;
        }
    }
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    sk_table_release(c);

    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)c->table_load_pct < (uint64_t)nrows * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    c->slots_log2 = lg;
    SK_HIP(c, hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    const size_t cbytes = (size_t)(nrows ? nrows : 1) * ncols * sizeof(uint32_t);
    SK_HIP(c, hipMalloc((void **)&c->d_counts, cbytes));
    SK_HIP(c, hipMemsetAsync(c->d_counts, 0, cbytes, c->stream));
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream));
    c->flags_ready = false;
    if (nrows) {
        uint64_t *d_in = NULL;
        SK_HIP(c, hipMalloc((void **)&d_in, (size_t)nrows * sizeof(uint64_t)));
        SK_HIP(c, hipMemcpyAsync(d_in, keys, (size_t)nrows * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        if (locality) {
            SK_HIP(c, hipMalloc((void **)&c->d_perm, (size_t)nrows * 4));
            SK_HIP(c, hipMalloc((void **)&c->d_inv, (size_t)nrows * 4));
            SK_HIP(c, hipMalloc((void **)&c->d_tmp, (size_t)nrows * 4));
            c->h_perm.resize(nrows);
            for (uint32_t i = 0; i < nrows; i++) c->h_perm[i] = locality[i] & 0x7FFFFFFFu;
            SK_HIP(c, hipMalloc((void **)&c->d_locality, (size_t)nrows * 4));
            SK_HIP(c, hipMemcpyAsync(c->d_locality, locality, (size_t)nrows * 4, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(sk_perm_from_locality, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_perm, (const uint32_t *)c->d_locality, nrows);   // (not a second 4 N bytes over PCIe)
            SK_HIP(c, hipMemsetAsync(c->d_inv, 0xFF, (size_t)nrows * 4, c->stream));
            hipLaunchKernelGGL(sk_invert_perm, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_inv, c->d_perm, nrows, c->d_flags);
        }
        hipLaunchKernelGGL(sk_table_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream,
                           d_in, nrows, c->d_keys, (uint32_t)(slots - 1), c->d_flags, c->d_perm, c->d_locality);
        {   // grid filters.  Level 1: 5.75 bits per key -- 3.4 MiB for a 5 Mbp strain, which stays in the L2 (4 MiB per XCD,
            // shared with everything else; round 3, once the kernel was bound by vector instructions: 3.0 -> 3.5 MiB saves more in
            // false positives' rounds than it costs in L2 misses: -1 % / -3 % / -6 % at 0 / 2 / 30 % strain reads; 4.5 MiB and up lose).  For bigger strains it is better to keep the 5 bits per key and leave the L2
            // than to keep the size and let the filter fill up (tools/grid_size_sweep.sh, 20 Mbp strain: 3 MiB 540,
            // 6 MiB 731, 12 MiB 850, 24 MiB 815 Gbase/s; 100 Mbp: 3 MiB 192, 64 MiB 465): misses of a sparse level 1
            // are served by the 256 MB Infinity Cache, the level-2 lookups a full one lets through are not.
            // Level 2 settles what level 1 lets through: >= 32 bits per key, false positives ~1e-5.
            uint64_t kib = c->grid_kib > 0 ? (uint64_t)c->grid_kib : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
            if (kib > (1ull << 22)) kib = 1ull << 22;
            if (kib < 4ull) kib = 4ull;
            c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
            uint32_t g2 = 12;
            while (g2 < 34 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
            c->grid2_blocks_log2 = g2 - 6u;
            const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
            SK_HIP(c, hipMalloc((void **)&c->d_grid1, b1));
            SK_HIP(c, hipMalloc((void **)&c->d_grid2, b2));
            SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
            SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
            // filled when it is known from what: by sk_table_load_text from the strain's text (one insert per base instead of
            // sixteen per key), or -- no text stage -- from the slots when the first scan asks (sk_grid_ensure)
            c->grid_pending = true;
            c->all_rows_in_text = false;
        }
        uint32_t flags[2] = {0, 0};
        SK_HIP(c, hipMemcpyAsync(flags, c->d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        SK_HIP(c, hipStreamSynchronize(c->stream));
        hipFree(d_in);
        if (flags[1]) { sk_table_release(c); return sk_fail(c, SK_E_DUPKEY, "%u duplicate/malformed keys", flags[1]); }
    }
    SK_HIP(c, hipGetLastError());
    c->nrows = nrows;
    c->ncols = ncols;
    return SK_OK;
}

extern "C" int sk_table_load_wide(sk_ctx *c, const char *keys31, const uint32_t *rows, uint32_t nwide)
{
    if (!c) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "sk_table_load first");
    if (nwide == 0) return SK_OK;
    if (!keys31 || !rows) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    uint32_t lg = 4;
    while (((uint32_t)1 << lg) < nwide * 2u) lg++;
    const uint32_t wslots = (uint32_t)1 << lg;
    std::vector<uint32_t> index(wslots, 0u);
    for (uint32_t i = 0; i < nwide; i++) {
        if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "wide row %u out of range", rows[i]);
        uint32_t slot = sk_hash_wide(keys31 + (size_t)i * 32) & (wslots - 1);
        while (index[slot]) slot = (slot + 1) & (wslots - 1);
        index[slot] = i + 1;
    }
    SK_HIP(c, hipMalloc((void **)&c->d_wide_keys, (size_t)nwide * 32));
    SK_HIP(c, hipMalloc((void **)&c->d_wide_rows, (size_t)nwide * sizeof(uint32_t)));
    SK_HIP(c, hipMalloc((void **)&c->d_wide_index, (size_t)wslots * sizeof(uint32_t)));
    SK_HIP(c, hipMemcpy(c->d_wide_keys, keys31, (size_t)nwide * 32, hipMemcpyHostToDevice));
    std::vector<uint32_t> wrows(rows, rows + nwide);
    if (!c->h_perm.empty()) for (uint32_t i = 0; i < nwide; i++) wrows[i] = c->h_perm[rows[i]];
    SK_HIP(c, hipMemcpy(c->d_wide_rows, wrows.data(), (size_t)nwide * sizeof(uint32_t), hipMemcpyHostToDevice));
    SK_HIP(c, hipMemcpy(c->d_wide_index, index.data(), (size_t)wslots * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->wide_mask = wslots - 1;
    c->nwide = nwide;
    return SK_OK;
}

extern "C" int sk_table_load_text(sk_ctx *c, const uint32_t *text2, uint32_t nbases, const uint32_t *first_pos)
{
    if (!c || !text2 || !first_pos) return SK_E_ARG;
    if (!c->d_keys || !c->nrows) return sk_fail(c, SK_E_STATE, "sk_table_load_ex first");
    if (c->h_perm.empty()) return sk_fail(c, SK_E_STATE, "the text stage needs the locality order of sk_table_load_ex");
    if (nbases < SK_K || nbases > 0x7FFFFF00u) return sk_fail(c, SK_E_ARG, "text of %u bases (a table slot holds 31 bits of position)", nbases);
    const uint32_t nrows = c->nrows;
    // by counter index; the contract: rows with a position first, positions ascending
    std::vector<uint32_t> pos_by_idx(nrows, 0xFFFFFFFFu);
    for (uint32_t r = 0; r < nrows; r++) pos_by_idx[c->h_perm[r]] = first_pos[r];
    uint32_t m = 0;
    while (m < nrows && pos_by_idx[m] != 0xFFFFFFFFu) m++;
    for (uint32_t i = 0; i < nrows; i++) {
        const uint32_t p = pos_by_idx[i];
        if (i >= m) { if (p != 0xFFFFFFFFu) return sk_fail(c, SK_E_ARG, "rows with a text position must come first in locality order"); continue; }
        if (p > nbases - SK_K || (i && p <= pos_by_idx[i - 1])) return sk_fail(c, SK_E_ARG, "text positions must ascend with the locality order and lie inside the text");
    }
    // rank map: per 64 positions {counter index of the first row starting in the block, the 64 start bits}
    const size_t nblk = (size_t)nbases / 64 + 2;
    std::vector<sk_u4> rank(nblk, (sk_u4){0u, 0u, 0u, 0u});
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t p = pos_by_idx[i];
        if (p & 32u) rank[p >> 6].z |= 1u << (p & 31u); else rank[p >> 6].y |= 1u << (p & 31u);
    }
    uint32_t run = 0;
    for (size_t b = 0; b < nblk; b++) { rank[b].x = run; run += (uint32_t)__builtin_popcount(rank[b].y) + (uint32_t)__builtin_popcount(rank[b].z); }
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(c->d_text2); c->d_text2 = NULL;
    hipFree(c->d_rank); c->d_rank = NULL;
    c->text_bases = 0;
    const size_t words = (size_t)nbases / 16 + 4, have = ((size_t)nbases + 15) / 16;
    SK_HIP(c, hipMalloc((void **)&c->d_text2, words * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    hipFree(c->d_diff); c->d_diff = NULL;
    hipFree(c->d_diff_sums); c->d_diff_sums = NULL;
    c->diff_col = -1;
    SK_HIP(c, hipMalloc((void **)&c->d_diff, ((size_t)nrows + 2) * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_diff_sums, ((size_t)nrows / SK_DIFF_PER_BLOCK + 2) * 4));
    SK_HIP(c, hipMemsetAsync(c->d_diff, 0, ((size_t)nrows + 2) * 4, c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_text2, text2, have * 4, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_rank, rank.data(), nblk * sizeof(sk_u4), hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_tmp, pos_by_idx.data(), (size_t)nrows * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(sk_table_setpos, dim3(4096), dim3(256), 0, c->stream, c->d_keys, (uint64_t)1 << c->slots_log2, c->d_tmp);
    c->all_rows_in_text = m == nrows;
    if (c->grid_pending && c->all_rows_in_text) {          // (rows without a place in the text -- a U in the strain -- : from the slots, sk_grid_ensure)
        hipLaunchKernelGGL(sk_grid_insert_text, dim3((nbases + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->d_text2, nbases,
                           (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        c->grid_pending = false;
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipGetLastError());
    c->text_bases = nbases;
    return SK_OK;
}

// The whole table from the strain's text, on the device (round 3; strain_detect's opening: src/strain_detect.c:137-139 builds it with
// GEN_hash_sequences_set_count_vec(r, 31, h, 1, 0, 0, 6), src/genome_compare.c:967-1030).  The host parses the file and hands over the
// bases (2 bits each, records end to end) and, per position, whether a window of 31 A/C/G/T bases of one record starts there; keys,
// first occurrences, row numbers (by first occurrence: "strain order"), rank map, both filter levels and column 0 are made here --
// no 40 MB of keys, no permutations and no column 0 over PCIe, no hash table on the host.  *nrows_out = distinct keys.
static int sk_table_build_from_text_steps(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                          uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out, uint32_t *&d_ok);
extern "C" int sk_table_build_from_text(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                        uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out)
{
    if (!c || !text2 || !startok || !nrows_out || ncols == 0 || ncols > 16) return SK_E_ARG;
    if (nbases < SK_K || nbases > 0x7FFFFF00u) return sk_fail(c, SK_E_ARG, "text of %u bases (a table slot holds 31 bits of position)", nbases);
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    sk_table_release(c);
    // a build that fails midway (out of memory with many strains on one device) must leave neither its scratch nor half a table
    // behind: the caller falls back to the host builder on this same context
    uint32_t *d_ok = NULL;
    const int rc = sk_table_build_from_text_steps(c, text2, startok, nbases, nstarts, ncols, col0_value, nrows_out, d_ok);
    if (d_ok) (void)hipFree(d_ok);
    if (rc != SK_OK) {
        (void)hipStreamSynchronize(c->stream);
        sk_table_release(c);
        c->nrows = 0; c->ncols = 0;
    }
    return rc;
}

static int sk_table_build_from_text_steps(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                          uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out, uint32_t *&d_ok)
{
    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)c->table_load_pct < (uint64_t)nstarts * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    const uint32_t mask = (uint32_t)(slots - 1);
    c->slots_log2 = lg;
    const size_t words = (size_t)nbases / 16 + 4, have = ((size_t)nbases + 15) / 16, nblk = (size_t)nbases / 64 + 2, bwords = ((size_t)nbases + 31) / 32;
    uint32_t *d_total = NULL;
    SK_HIP(c, hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    SK_HIP(c, hipMalloc((void **)&c->d_text2, words * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    SK_HIP(c, hipMalloc((void **)&d_ok, bwords * 4 + 4));
    d_total = d_ok + bwords;
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    SK_HIP(c, hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_rank, 0, nblk * sizeof(sk_u4), c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream));
    c->flags_ready = false;
    SK_HIP(c, hipMemcpyAsync(c->d_text2, text2, have * 4, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(d_ok, startok, bwords * 4, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((nbases + 255) / 256), block(256);
    hipLaunchKernelGGL(sk_build_insert, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, (const uint32_t *)d_ok, nbases, c->d_keys, mask);
    hipLaunchKernelGGL(sk_build_first, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, (const uint32_t *)d_ok, nbases, (const sk_u4 *)c->d_keys, mask, c->d_rank);
    hipLaunchKernelGGL(sk_build_rank_scan, dim3(1), dim3(1024), 0, c->stream, c->d_rank, (uint32_t)nblk, d_total);
    uint32_t nrows = 0;
    SK_HIP(c, hipMemcpyAsync(&nrows, d_total, 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_ok);
    d_ok = NULL;
    *nrows_out = nrows;
    c->nrows = nrows; c->ncols = ncols;
    c->text_bases = nbases;
    c->all_rows_in_text = true;
    const size_t cbytes = (size_t)(nrows ? nrows : 1) * ncols * sizeof(uint32_t);
    SK_HIP(c, hipMalloc((void **)&c->d_counts, cbytes));
    SK_HIP(c, hipMemsetAsync(c->d_counts, 0, cbytes, c->stream));
    if (nrows) {
        SK_HIP(c, hipMalloc((void **)&c->d_keys_by_row, (size_t)nrows * 8));
        hipLaunchKernelGGL(sk_build_index, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, nbases, c->d_keys, mask, (const sk_u4 *)c->d_rank, c->d_keys_by_row);
        if (col0_value) hipLaunchKernelGGL(sk_fill32, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_counts, nrows, col0_value);
        // the filter levels, sized as sk_table_load_ex sizes them, filled from the text
        uint64_t kib = c->grid_kib > 0 ? (uint64_t)c->grid_kib : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
        if (kib > (1ull << 22)) kib = 1ull << 22;
        if (kib < 4ull) kib = 4ull;
        c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
        uint32_t g2 = 12;
        while (g2 < 34 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
        c->grid2_blocks_log2 = g2 - 6u;
        const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
        SK_HIP(c, hipMalloc((void **)&c->d_grid1, b1));
        SK_HIP(c, hipMalloc((void **)&c->d_grid2, b2));
        SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
        SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
        hipLaunchKernelGGL(sk_grid_insert_text, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, nbases,
                           (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        c->grid_pending = false;
        c->diff_col = -1;
        SK_HIP(c, hipMalloc((void **)&c->d_diff, ((size_t)nrows + 2) * 4));
        SK_HIP(c, hipMalloc((void **)&c->d_diff_sums, ((size_t)nrows / SK_DIFF_PER_BLOCK + 2) * 4));
        SK_HIP(c, hipMemsetAsync(c->d_diff, 0, ((size_t)nrows + 2) * 4, c->stream));
    } else {                                             // (no key at all: an empty table the scans skip)
        c->grid1_blocks = 512; c->grid2_blocks_log2 = 6;
        SK_HIP(c, hipMalloc((void **)&c->d_grid1, 512 * sizeof(uint2)));
        SK_HIP(c, hipMalloc((void **)&c->d_grid2, 64 * sizeof(uint2)));
        SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, 512 * sizeof(uint2), c->stream));
        SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, 64 * sizeof(uint2), c->stream));
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

// the keys of a table built by sk_table_build_from_text, in row order (the device keeps its list: 8 bytes per row)
extern "C" int sk_table_export_keys(sk_ctx *c, uint64_t *keys_out)
{
    if (!c || (!keys_out && c->nrows)) return SK_E_ARG;
    if (!c->d_keys_by_row && c->nrows) return sk_fail(c, SK_E_STATE, "no key list to export (sk_table_build_from_text first)");
    SK_HIP(c, hipSetDevice(c->device));
    if (c->nrows) SK_HIP(c, hipMemcpy(keys_out, c->d_keys_by_row, (size_t)c->nrows * 8, hipMemcpyDeviceToHost));
    return SK_OK;
}
// ... of n chosen rows only (strain_detect prints the k-mers of informative rows -- 1 % of them -- and of no others: 40 MB per
// strain stay where they are)
extern "C" int sk_table_export_keys_of(sk_ctx *c, const uint32_t *rows, uint32_t n, uint64_t *keys_out)
{
    if (!c || (n && (!rows || !keys_out))) return SK_E_ARG;
    if (!n) return SK_OK;
    if (!c->d_keys_by_row) return sk_fail(c, SK_E_STATE, "no key list to export (sk_table_build_from_text first)");
    for (uint32_t i = 0; i < n; i++) if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "row %u out of range", rows[i]);
    SK_HIP(c, hipSetDevice(c->device));
    void *d = NULL;
    SK_HIP(c, hipMalloc(&d, (size_t)n * 12));
    uint64_t *d_out = (uint64_t *)d;
    uint32_t *d_rows = (uint32_t *)(d_out + n);
    hipError_t e = hipMemcpyAsync(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sk_gather_keys, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_out, (const uint64_t *)c->d_keys_by_row, (const uint32_t *)d_rows, n);
        e = hipMemcpyAsync(keys_out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    SK_HIP(c, e);
    return SK_OK;
}

// the filter levels of a table that has no text stage (or keys outside it): from the slots, once, before the first scan
static int sk_grid_ensure(sk_ctx *c)
{
    if (!c->grid_pending) return SK_OK;
    hipLaunchKernelGGL(sk_grid_insert_slots, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)c->d_keys, (uint64_t)1 << c->slots_log2,
                       (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
    SK_HIP(c, hipGetLastError());
    c->grid_pending = false;
    return SK_OK;
}

// Fold the pending difference array into its column (asynchronous on the context's stream).  Every entry point that
// reads or writes the counters calls this first; scans into another column do too.
static int sk_diff_flush(sk_ctx *c)
{
    if (c->diff_col < 0 || !c->d_diff) { c->diff_col = -1; return SK_OK; }
    const uint32_t n = c->nrows + 1u, nb = (n + SK_DIFF_PER_BLOCK - 1u) / SK_DIFF_PER_BLOCK;
    hipLaunchKernelGGL(sk_diff_block_sums, dim3(nb), dim3(256), 0, c->stream, c->d_diff, n, c->d_diff_sums);
    hipLaunchKernelGGL(sk_diff_scan_sums, dim3(1), dim3(1024), 0, c->stream, c->d_diff_sums, nb);
    hipLaunchKernelGGL(sk_diff_apply, dim3(nb), dim3(256), 0, c->stream, c->d_diff, n, c->d_diff_sums,
                       c->d_counts + (size_t)c->diff_col * c->nrows, c->nrows);
    c->diff_col = -1;
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

static int sk_scratch(sk_ctx *c, void **p, size_t *cap, size_t need);

// launch main + wide kernels over one device-resident batch
static int sk_launch_scan(sk_ctx *c, const uint8_t *d_stream, uint64_t nbytes, uint64_t emit_begin, uint32_t col,
                          const sk_sink *tally_sink = NULL)
{
    if (nbytes <= emit_begin || c->nrows == 0) return SK_OK;        // (an empty key set: nothing can be counted)
    const uint64_t ntiles = (nbytes + SK_TILE - 1) / SK_TILE;
    if (ntiles > 0x7FFFFFFFull) return sk_fail(c, SK_E_ARG, "batch too large");
    sk_table_view tv;
    tv.nrows = c->nrows;
    tv.text2 = c->no_text ? NULL : c->d_text2; tv.rank = c->d_rank; tv.text_bases = c->text_bases;
    tv.slots = c->d_keys; tv.mask = (uint32_t)(((uint64_t)1 << c->slots_log2) - 1);
    tv.oddlist = c->d_oddlist; tv.oddcap = c->odd_cap ? (uint32_t)c->odd_cap : SK_ODDCAP;
    tv.grid1 = c->d_grid1; tv.grid2 = c->d_grid2;
    tv.grid1_blocks = c->grid1_blocks; tv.grid2_shift = 32u - c->grid2_blocks_log2;
    sk_wide_view wv;
    wv.keys31 = c->d_wide_keys; wv.rows = c->d_wide_rows; wv.index = c->d_wide_index;
    wv.wmask = c->wide_mask; wv.nwide = c->nwide;
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    if (tally_sink) sink = *tally_sink;
    else {
        c->infbits_ok = false;                                 // (a column is about to change)
        sink.counts = c->d_counts + (size_t)col * c->nrows;
        sink.diff = c->d_diff;
        if (tv.text2 && c->diff_col != (int)col) {            // the difference array serves one column at a time
            int rc = sk_diff_flush(c);
            if (rc) return rc;
            c->diff_col = (int)col;
        }
    }
    const dim3 grid((uint32_t)ntiles), block(SK_THREADS);
    if (!c->d_grid1) return sk_fail(c, SK_E_STATE, "no table loaded");
    { int grc = sk_grid_ensure(c); if (grc) return grc; }

    // [0] odd bytes seen, [2] listed chunks: two sets of four words taking turns -- this launch's byte-string kernel zeroes the other set
    // for the next launch (both sets are zero after a table load)
    if (!c->flags_ready) {                                    // (the first scan after a table was built: its kernels used the same words)
        SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 8 * sizeof(uint32_t), c->stream));
        c->flags_ready = true;
    }
    c->flag_set ^= 1u;
    uint32_t *const d_fl = c->d_flags + 4u * c->flag_set, *const d_fl_next = c->d_flags + 4u * (c->flag_set ^ 1u);
    // timing: a begin/end event pair per launch from a small ring -- the oldest pair is added to the totals (its launch is
    // long over, SK_EV_PAIRS launches later) and used again, so a program that never asks for the timing holds 128 events,
    // not two per launch
    hipEvent_t e0 = NULL, e1 = NULL;
    const bool timed = true;
    if (c->ev.size() >= 2 * SK_EV_PAIRS) {
        float ms = 0.f;
        SK_HIP(c, hipEventSynchronize(c->ev[1]));
        SK_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->timed_ms += ms;
        c->timed_launches++;
        c->ev_free.push_back(c->ev[0]);
        c->ev_free.push_back(c->ev[1]);
        c->ev.erase(c->ev.begin(), c->ev.begin() + 2);
    }
    if (c->ev_free.size() >= 2) {
        e1 = c->ev_free.back(); c->ev_free.pop_back();
        e0 = c->ev_free.back(); c->ev_free.pop_back();
    } else {
        SK_HIP(c, hipEventCreate(&e0));
        SK_HIP(c, hipEventCreate(&e1));
    }
    SK_HIP(c, hipEventRecord(e0, c->stream));
    // The partitioned pipeline (the level-1 question answered from LDS) is an experiment kept selectable (option
    // "pipeline" = 2; parity-tested): as measured it LOSES to the single kernel (0.43 against 0.34 ms per 0.6 Gbase at
    // 2 % strain reads; profiles/r02_lds_pipeline.txt) -- its one full pass over the stream plus the
    // bin write already costs half of the single kernel's time, and the re-read of the candidates' neighbourhoods pays
    // the fabric's random-line rate.  The default is the single kernel for every batch size.
    const bool piped = (!c->ablate || c->ablate >= 7) && c->pipeline == 2;        // (ablations 7-9 exist for both forms)
    const uint8_t *d_cand = NULL;
    if (piped && !c->d_grid3) {
        // the partitioned pipeline's filter slices, built from the resident table the first time they are wanted:
        // SK_BIN_P bitmaps of 2^20 bits (16 MiB)
        const size_t b3 = (size_t)SK_BIN_P * SK_BIN_WORDS * 4;
        const uint64_t nslots = (uint64_t)1 << c->slots_log2;
        SK_HIP(c, hipMalloc((void **)&c->d_grid3, b3));
        SK_HIP(c, hipMemsetAsync(c->d_grid3, 0, b3, c->stream));
        hipLaunchKernelGGL(sk_grid3_insert, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, c->stream, (const sk_u4 *)c->d_keys, nslots, c->d_grid3);
    }
    if (piped) {
        const uint64_t ntiles_bin = (nbytes + SK_BIN_TILE - 1) / SK_BIN_TILE;
        int rc;
        if ((rc = sk_scratch(c, &c->p_bins, &c->p_bins_cap, (size_t)ntiles_bin * SK_BIN_P * SK_BIN_CAP * 4)) != SK_OK) return rc;
        if ((rc = sk_scratch(c, &c->p_binn, &c->p_binn_cap, (size_t)ntiles_bin * SK_BIN_P)) != SK_OK) return rc;
        if ((rc = sk_scratch(c, &c->p_cand, &c->p_cand_cap, (size_t)ntiles_bin * SK_BIN_CH + 64)) != SK_OK) return rc;
        SK_HIP(c, hipMemsetAsync(c->p_cand, 0, (size_t)ntiles_bin * SK_BIN_CH + 64, c->stream));
        hipLaunchKernelGGL(sk_bin, dim3((uint32_t)ntiles_bin), dim3(256), 0, c->stream, d_stream, nbytes, tv,
                           (uint32_t *)c->p_bins, (uint8_t *)c->p_binn, (uint32_t)ntiles_bin, (uint8_t *)c->p_cand, d_fl);
        // one workgroup per CU at a time (128 KiB of LDS each): a few shares per partition keep all 256 CUs busy
        uint32_t splits = ntiles_bin >= 4096 ? 4u : ntiles_bin >= 1024 ? 2u : 1u;
        static bool lds_attr = false;
        if (!lds_attr) {
            SK_HIP(c, hipFuncSetAttribute((const void *)sk_lds_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SK_BIN_WORDS * 4)));
            lds_attr = true;
        }
        hipLaunchKernelGGL(sk_lds_probe, dim3(SK_BIN_P * splits), dim3(1024), SK_BIN_WORDS * 4, c->stream, (const uint32_t *)c->d_grid3,
                           (const uint32_t *)c->p_bins, (const uint8_t *)c->p_binn, (uint32_t)ntiles_bin, splits, (uint8_t *)c->p_cand);
        d_cand = (const uint8_t *)c->p_cand;
    }
#define SK_LAUNCH_GRID(T, A, C) hipLaunchKernelGGL((sk_scan_grid<T, A, C>), grid, block, 0, c->stream, \
                                                   d_stream, nbytes, emit_begin, tv, sink, d_fl, d_cand)
    if (tally_sink && tally_sink->ns)
        hipLaunchKernelGGL((sk_scan_grid<true, 0, false, true>), grid, block, 0, c->stream, d_stream, nbytes, emit_begin, tv, sink, d_fl, d_cand);
    else if (piped && tally_sink) SK_LAUNCH_GRID(true, 0, true);
#ifdef SK_EXPERIMENTS
    else if (piped && c->ablate == 7) SK_LAUNCH_GRID(false, 7, true);
    else if (piped && c->ablate == 8) SK_LAUNCH_GRID(false, 8, true);
    else if (piped && c->ablate == 9) SK_LAUNCH_GRID(false, 9, true);
#endif
    else if (piped)          SK_LAUNCH_GRID(false, 0, true);
    else if (tally_sink)     SK_LAUNCH_GRID(true, 0, false);
#ifdef SK_EXPERIMENTS
    else if (c->ablate == 1) SK_LAUNCH_GRID(false, 1, false);
    else if (c->ablate == 2) SK_LAUNCH_GRID(false, 2, false);
    else if (c->ablate == 3) SK_LAUNCH_GRID(false, 3, false);
    else if (c->ablate == 4) SK_LAUNCH_GRID(false, 4, false);
    else if (c->ablate == 5) SK_LAUNCH_GRID(false, 5, false);
    else if (c->ablate == 6) SK_LAUNCH_GRID(false, 6, false);
    else if (c->ablate == 10) SK_LAUNCH_GRID(false, 10, false);
#endif
    else                     SK_LAUNCH_GRID(false, 0, false);
#undef SK_LAUNCH_GRID
    if (timed) {
        SK_HIP(c, hipEventRecord(e1, c->stream));
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
    uint64_t wblocks = (nbytes - emit_begin + 255) / 256;
    if (wblocks > 2048) wblocks = 2048;                        // (grid-stride: eight workgroups per CU; with nothing to do -- the usual case -- the launch is over in 3 us)
    if (wblocks == 0) { wblocks = 1; }                        // (the kernel also readies the next launch's flag words)
    if (tally_sink && tally_sink->ns)
        hipLaunchKernelGGL((sk_scan_wide<true, true>), dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    else if (tally_sink)
        hipLaunchKernelGGL(sk_scan_wide<true>, dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    else
        hipLaunchKernelGGL(sk_scan_wide<false>, dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

// grow-only scratch buffer of the context
static int sk_scratch(sk_ctx *c, void **p, size_t *cap, size_t need)
{
    if (need <= *cap) return SK_OK;
    if (*p) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(*p); *p = NULL; *cap = 0; }
    size_t want = need + need / 4 + 4096;
    SK_HIP(c, hipMalloc(p, want));
    *cap = want;
    return SK_OK;
}

// ---------------------------------------------------------------------------------------------
// strain_detect: a batch of records resident on the device, tallied against any number of tables
// ---------------------------------------------------------------------------------------------
struct sk_batch {
    sk_ctx      *owner;
    hipStream_t  stream;
    hipEvent_t   ready;                 // the upload of the current contents
    void        *d_stream, *d_rec;      // bytes; rec_start[nrec] followed by tile_first[ntiles + 2]
    size_t       stream_cap, rec_cap;
    uint64_t     nbytes;
    uint32_t     nrec, ntiles;
    std::vector<uint32_t> tile_first;
};

extern "C" int sk_batch_create(sk_ctx *c, sk_batch **out)
{
    if (!c || !out) return SK_E_ARG;
    *out = NULL;
    SK_HIP(c, hipSetDevice(c->device));
    sk_batch *b = new (std::nothrow) sk_batch();
    if (!b) return SK_E_NOMEM;
    b->owner = c;
    b->d_stream = b->d_rec = NULL; b->stream_cap = b->rec_cap = 0; b->nbytes = 0; b->nrec = b->ntiles = 0;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { delete b; return sk_fail(c, SK_E_HIP, "stream"); }
    if (hipEventCreateWithFlags(&b->ready, hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(b->stream); delete b; return sk_fail(c, SK_E_HIP, "event"); }
    *out = b;
    return SK_OK;
}

// waits until the batch's uploads are done: its device buffers may be refilled and the host memory it was filled from reused
extern "C" int sk_batch_sync(sk_batch *b)
{
    if (!b) return SK_E_ARG;
    SK_HIP(b->owner, hipSetDevice(b->owner->device));
    SK_HIP(b->owner, hipStreamSynchronize(b->stream));
    return SK_OK;
}

extern "C" void sk_batch_destroy(sk_batch *b)
{
    if (!b) return;
    hipSetDevice(b->owner->device);
    hipStreamSynchronize(b->stream);
    hipFree(b->d_stream); hipFree(b->d_rec);
    hipEventDestroy(b->ready);
    hipStreamDestroy(b->stream);
    delete b;
}

// Upload new contents.  Every tally launched on the previous contents must have been collected.
extern "C" int sk_batch_fill(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec)
{
    if (!b || !stream || !rec_start) return SK_E_ARG;
    sk_ctx *c = b->owner;
    if (nbytes == 0 || nrec == 0 || nbytes > 0xFFFFFFF0ull) return sk_fail(c, SK_E_ARG, "bad batch size");
    SK_HIP(c, hipSetDevice(c->device));
    const uint32_t ntiles = (uint32_t)((nbytes + 32767u) >> 15);
    if (nbytes + 16 > b->stream_cap) {
        SK_HIP(c, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_stream); b->d_stream = NULL; b->stream_cap = 0;
        const size_t want = nbytes + nbytes / 4 + 4096;
        SK_HIP(c, hipMalloc(&b->d_stream, want));
        b->stream_cap = want;
    }
    const size_t rec_need = ((size_t)nrec + ntiles + 2) * 4;
    if (rec_need > b->rec_cap) {
        SK_HIP(c, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_rec); b->d_rec = NULL; b->rec_cap = 0;
        const size_t want = rec_need + rec_need / 4 + 4096;
        SK_HIP(c, hipMalloc(&b->d_rec, want));
        b->rec_cap = want;
    }
    SK_HIP(c, hipStreamSynchronize(b->stream));            // tile_first (host vector) of the previous upload is free again
    b->tile_first.resize((size_t)ntiles + 2);
    for (uint32_t t = 0, r = 0; t < ntiles + 2; t++) {     // first record starting at or after the tile's first byte
        const uint64_t edge = (uint64_t)t << 15;
        while (r < nrec && rec_start[r] < edge) r++;
        b->tile_first[t] = r;
    }
    SK_HIP(c, hipMemcpyAsync(b->d_stream, stream, nbytes, hipMemcpyHostToDevice, b->stream));
    SK_HIP(c, hipMemcpyAsync(b->d_rec, rec_start, (size_t)nrec * 4, hipMemcpyHostToDevice, b->stream));
    SK_HIP(c, hipMemcpyAsync((uint32_t *)b->d_rec + nrec, b->tile_first.data(), (size_t)(ntiles + 2) * 4, hipMemcpyHostToDevice, b->stream));
    SK_HIP(c, hipEventRecord(b->ready, b->stream));
    b->nbytes = nbytes; b->nrec = nrec; b->ntiles = ntiles;
    return SK_OK;
}

// Start the tallies of batch `b` against the table of context `c` (any context on the batch's device);
// returns at once.  One launch per context may be in flight.
extern "C" int sk_tally_launch(sk_ctx *c, const sk_batch *b, uint32_t type_col, uint32_t informative_value, uint64_t hits_cap)
{
    if (!c || !b) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (type_col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", type_col);
    if (b->owner->device != c->device) return sk_fail(c, SK_E_ARG, "batch lives on another device");
    if (b->nrec == 0) return sk_fail(c, SK_E_STATE, "empty batch");
    SK_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = sk_diff_flush(c)) != SK_OK) return rc;              // (the type column is read by the kernel)
    const uint32_t nrec = b->nrec;
    if ((rc = sk_scratch(c, &c->t_tally, &c->t_tally_cap, (size_t)nrec * 8 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_compact, &c->t_compact_cap, (size_t)nrec * 12 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_hits, &c->t_hits_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    if ((size_t)nrec * 8 + 16 > c->h_tally_cap) {
        if (c->h_tally) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_tally); c->h_tally = NULL; c->h_tally_cap = 0; }
        const size_t want = (size_t)nrec * 10 + 4096;
        SK_HIP(c, hipHostMalloc((void **)&c->h_tally, want, hipHostMallocDefault));
        c->h_tally_cap = want;
    }
    unsigned long long *d_n = (unsigned long long *)((uint8_t *)c->t_tally + (size_t)nrec * 8);   // hit counter behind the tallies
    SK_HIP(c, hipStreamWaitEvent(c->stream, b->ready, 0));
    SK_HIP(c, hipMemsetAsync(c->t_tally, 0, (size_t)nrec * 8 + 16, c->stream));
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    sink.rec_start = (const uint32_t *)b->d_rec; sink.nrec = nrec; sink.tally = (uint32_t *)c->t_tally;
    sink.tile_first = (const uint32_t *)b->d_rec + nrec;
    sink.type = c->d_counts + (size_t)type_col * c->nrows; sink.inf_value = informative_value;
    sink.hits = (uint2 *)c->t_hits; sink.nhits = d_n; sink.hits_cap = hits_cap; sink.inv = c->d_inv;
    if (!c->infbits_ok || c->infbits_col != type_col || c->infbits_val != informative_value) {
        if (!c->d_infbits) SK_HIP(c, hipMalloc((void **)&c->d_infbits, ((size_t)c->nrows / 32 + 8) * 4));
        SK_HIP(c, hipMemsetAsync(c->d_infbits, 0, ((size_t)c->nrows / 32 + 8) * 4, c->stream));
        hipLaunchKernelGGL(sk_inf_bitmap, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, sink.type, c->nrows, informative_value, c->d_infbits);
        c->infbits_ok = true; c->infbits_col = type_col; c->infbits_val = informative_value;
    }
    sink.infbits = c->d_infbits;
    rc = sk_launch_scan(c, (const uint8_t *)b->d_stream, b->nbytes, 0, 0, &sink);
    if (rc) return rc;
    // the records that were hit at all, compacted on the device (sk_tally_collect_sparse); only the two counters come
    // back now, the tallies themselves when they are asked for
    hipLaunchKernelGGL(sk_tally_compact, dim3((nrec + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->t_tally, nrec,
                       (uint32_t *)c->t_compact, d_n + 1);
    SK_HIP(c, hipMemcpyAsync(c->h_tally, d_n, 16, hipMemcpyDeviceToHost, c->stream));
    c->t_inflight_nrec = nrec;
    c->t_inflight_cap = hits_cap;
    return SK_OK;
}

// Wait for the launch of this context and hand out its results.  *out_nhits may exceed hits_cap (log
// overflow): launch again with more room.
extern "C" int sk_tally_collect(sk_ctx *c, uint32_t *out_tally, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!c || !out_tally || !out_nhits) return SK_E_ARG;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t nrec = c->t_inflight_nrec;
    c->t_inflight_nrec = 0;
    SK_HIP(c, hipMemcpy(out_tally, c->t_tally, (size_t)nrec * 8, hipMemcpyDeviceToHost));
    unsigned long long nh;
    memcpy(&nh, c->h_tally, 8);
    const unsigned long long take = nh < c->t_inflight_cap ? nh : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    if (take) SK_HIP(c, hipMemcpy(out_hits, c->t_hits, (size_t)take * sizeof(uint2), hipMemcpyDeviceToHost));
    *out_nhits = nh;
    return SK_OK;
}

// Like sk_tally_collect, but only the records with at least one hit come back: out[i] = {record, all hits, informative
// hits}, unordered, *n of them (if *n > cap only cap were stored: collect again is not possible -- size cap to the batch's
// record count to be safe, or to what the workload allows).
extern "C" int sk_tally_collect_sparse(sk_ctx *c, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!c || !n || !out_nhits || (cap && !out)) return SK_E_ARG;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    unsigned long long two[2];
    memcpy(two, c->h_tally, 16);
    const unsigned long long take_r = two[1] < cap ? two[1] : cap;
    if (take_r) SK_HIP(c, hipMemcpy(out, c->t_compact, (size_t)take_r * sizeof(sk_tally_rec), hipMemcpyDeviceToHost));
    *n = two[1];
    const unsigned long long take = two[0] < c->t_inflight_cap ? two[0] : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    if (take) SK_HIP(c, hipMemcpy(out_hits, c->t_hits, (size_t)take * sizeof(uint2), hipMemcpyDeviceToHost));
    *out_nhits = two[0];
    return SK_OK;
}

// Per-record tallies of one batch, synchronous: upload + launch + collect on the context's own batch.
extern "C" int sk_tally_batch(sk_ctx *c, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec,
                              uint32_t type_col, uint32_t informative_value, uint32_t *out_tally,
                              sk_hit *out_hits, uint64_t hits_cap, uint64_t *out_nhits)
{
    if (!c || !stream || !rec_start || !out_tally || !out_nhits || (hits_cap && !out_hits)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    int rc;
    if (!c->own_batch && (rc = sk_batch_create(c, &c->own_batch)) != SK_OK) return rc;
    if ((rc = sk_batch_fill(c->own_batch, stream, nbytes, rec_start, nrec)) != SK_OK) return rc;
    if ((rc = sk_tally_launch(c, c->own_batch, type_col, informative_value, hits_cap)) != SK_OK) return rc;
    return sk_tally_collect(c, out_tally, out_hits, out_nhits);
}


// ---------------------------------------------------------------------------------------------
// One table for several resident strains (strain_detect -S, BASELINE configs[4]: 32 strains per GPU against one
// metagenome).  Without it a batch of reads is scanned once per strain; with it once.  New relative to the reference,
// which holds one strain per process (src/strain_detect.c:137-146).
//
// The union is built on the device from the members' resident tables: global row g = base[s] + the member's counter
// index; the members' texts and rank maps are laid one behind the other (each padded to a multiple of 64 bases), so
// that seed-and-verify works on the union as it does on one strain.  A key that several strains hold gets ONE slot (the
// first member's: its row and text position), but every one of its global rows carries the same pair of masks
// {members that hold the key, members in which it is informative}: whichever row a hit comes out at -- the slot's, or
// the rank of a position in another member's text -- it names the same strains.  The hit log names, per informative
// strain, that member's own row (sk_union_resolve looks the key up in the member's table).
// ---------------------------------------------------------------------------------------------
struct sk_union_member { const sk_u4 *slots; uint32_t mask; const uint32_t *inv; };

__global__ void sk_union_insert(const sk_u4 *__restrict__ mslots, uint64_t nslots, uint32_t base, uint32_t tbase,
                                sk_u4 *slots, uint32_t mask, uint64_t *__restrict__ ukeys)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = mslots[i];
        const uint64_t k = sk_slot_key(e);
        if (k == SK_EMPTY64) continue;
        const uint32_t g = base + e.z;
        ukeys[g] = k;
        const uint32_t tp = e.w >> 1;
        const uint32_t w = tp == 0x7FFFFFFFu ? e.w : (((tbase + tp) << 1) | (e.w & 1u));
        uint32_t slot = sk_slot0(sk_khash(k), mask);
        for (;;) {
            const unsigned long long old = atomicCAS((unsigned long long *)&slots[slot], (unsigned long long)SK_EMPTY64, (unsigned long long)k);
            if (old == SK_EMPTY64) { ((uint32_t *)&slots[slot])[2] = g; ((uint32_t *)&slots[slot])[3] = w; break; }
            if (old == k) break;                           // an earlier member holds the key: its slot stands
            slot = (slot + 1u) & mask;
        }
    }
}

// pass A: every row finds its key's slot row ("canon") and sets its member's bits there
__global__ void sk_union_mask_a(const sk_u4 *__restrict__ mslots, uint64_t nslots, uint32_t base, uint32_t s,
                                const uint32_t *__restrict__ type, uint32_t inf_value,
                                const sk_u4 *__restrict__ slots, uint32_t mask, uint2 *umask, uint32_t *__restrict__ canon)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += stride) {
        const sk_u4 e = mslots[i];
        const uint64_t k = sk_slot_key(e);
        if (k == SK_EMPTY64) continue;
        uint32_t slot = sk_slot0(sk_khash(k), mask), w = 0xFFFFFFFFu;
        for (;;) {
            const sk_u4 u = slots[slot];
            const uint64_t uk = sk_slot_key(u);
            if (uk == k) { w = u.z; break; }
            if (uk == SK_EMPTY64) break;                   // (cannot happen: every key was inserted)
            slot = (slot + 1u) & mask;
        }
        const uint32_t g = base + e.z;
        canon[g] = w == 0xFFFFFFFFu ? g : w;
        if (w == 0xFFFFFFFFu) continue;
        atomicOr(&umask[w].x, 1u << s);
        if (type[e.z] == inf_value) atomicOr(&umask[w].y, 1u << s);
    }
}

// pass B: the other rows of a key copy the pair
__global__ void sk_union_mask_b(uint2 *umask, const uint32_t *__restrict__ canon, uint32_t n)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t w = canon[g];
    if (w != g && w < n) umask[g] = umask[w];
}

// a member's rank map into the union's: counter indices move by the member's base
__global__ void sk_union_rank_copy(sk_u4 *__restrict__ dst, const sk_u4 *__restrict__ src, uint32_t nsrc, uint32_t ndst, uint32_t base, uint32_t after)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ndst) return;
    if (b < nsrc) { sk_u4 r = src[b]; r.x += base; dst[b] = r; }
    else dst[b] = (sk_u4){after, 0u, 0u, 0u};             // padding: no row starts here
}

// The scan logged (position, global row) once per informative hit.  Dealt out here: one entry per strain in which the row's key
// is informative, {position, strain << SK_UNION_ROW_BITS | that member's OWN row of the key} (a probe of the key in the member's
// table).  *nout counts every entry, stored or not (the caller asks again with more room if it exceeds cap_out).
__global__ void sk_union_resolve(const uint2 *__restrict__ raw, const unsigned long long *__restrict__ nraw, unsigned long long cap_raw,
                                 uint2 *__restrict__ out, unsigned long long *nout, unsigned long long cap_out,
                                 const uint64_t *__restrict__ ukeys, const uint2 *__restrict__ umask, const sk_union_member *__restrict__ mem)
{
    const unsigned long long n = *nraw < cap_raw ? *nraw : cap_raw;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint2 e = raw[i];
        uint32_t infm = umask[e.y].y;
        const uint64_t k = ukeys[e.y];
        unsigned long long at = atomicAdd(nout, (unsigned long long)__popc(infm));
        while (infm) {
            const uint32_t s = (uint32_t)__builtin_ctz(infm);
            infm &= infm - 1u;
            const sk_union_member m = mem[s];
            uint32_t slot = sk_slot0(sk_khash(k), m.mask), row = (1u << SK_UNION_ROW_BITS) - 1u;
            for (;;) {
                const sk_u4 u = m.slots[slot];
                const uint64_t uk = sk_slot_key(u);
                if (uk == k) { row = m.inv ? m.inv[u.z] : u.z; break; }
                if (uk == SK_EMPTY64) break;
                slot = (slot + 1u) & m.mask;
            }
            if (at < cap_out) out[at] = make_uint2(e.x, (s << SK_UNION_ROW_BITS) | row);
            at++;
        }
    }
}

// The (record, strain) tallies of a union scan live in a dense array that is ALL ZERO between launches: the scan marks the
// records it touched (one byte each), and this pass visits only those -- appends {record * ns + strain, all, informative} for
// the pairs that were hit, zeroes them again and clears the mark.  Its cost follows the reads that hit a strain, not
// records x strains (57 MB per 32 MiB batch of 150-base reads and 32 strains; zeroing and sweeping that much took longer
// than the scan itself).
__global__ void sk_union_compact(uint2 *__restrict__ tally, uint8_t *__restrict__ flag, uint32_t nrec, uint32_t ns,
                                 uint32_t *__restrict__ out, unsigned long long *n)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec || !flag[r]) return;
    flag[r] = 0;
    uint2 *row = tally + (size_t)r * ns;
    uint32_t cnt = 0;
    for (uint32_t s = 0; s < ns; s++) cnt += (uint32_t)((row[s].x | row[s].y) != 0u);
    if (!cnt) return;
    unsigned long long at = atomicAdd(n, (unsigned long long)cnt);
    for (uint32_t s = 0; s < ns; s++) {
        const uint2 t = row[s];
        if (!(t.x | t.y)) continue;
        out[3 * at] = r * ns + s; out[3 * at + 1] = t.x; out[3 * at + 2] = t.y;
        at++;
        row[s] = make_uint2(0u, 0u);
    }
}

struct sk_union {
    sk_ctx   *uc;                    // the union as a context of its own (table, text, filters, scratch, stream)
    uint32_t  n;
    void     *d_tally, *d_flag, *d_raw;       // dense (record, strain) tallies (all zero between launches), touched marks, raw hit log
    size_t    tally_cap, flag_cap, raw_cap;
    unsigned long long *d_cnt;       // [0] raw log entries, [1] compacted pairs, [2] dealt-out log entries
    uint64_t *d_ukeys;               // [rows] key of every global row
    uint2    *d_umask;               // [rows]
    sk_union_member *d_members;
};

extern "C" void sk_union_destroy(sk_union *u)
{
    if (!u) return;
    if (u->uc) {
        hipSetDevice(u->uc->device);
        hipStreamSynchronize(u->uc->stream);
        hipFree(u->d_ukeys); hipFree(u->d_umask); hipFree(u->d_members);
        hipFree(u->d_tally); hipFree(u->d_flag); hipFree(u->d_raw); hipFree(u->d_cnt);
        sk_ctx_destroy(u->uc);
    }
    delete u;
}

extern "C" int sk_union_create(sk_ctx *const *members, uint32_t n, uint32_t type_col, uint32_t informative_value, sk_union **out)
{
    if (!members || !out || n < 1 || n > SK_UNION_MAX) return SK_E_ARG;
    *out = NULL;
    sk_ctx *first = members[0];
    if (!first) return SK_E_ARG;
    uint64_t rows = 0, tbases = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        if (!m) return SK_E_ARG;
        if (m->device != first->device) return sk_fail(first, SK_E_ARG, "the members of a union live on one device");
        if (!m->d_keys || !m->nrows) return sk_fail(first, SK_E_STATE, "member %u has no table", s);
        if (type_col >= m->ncols) return sk_fail(first, SK_E_ARG, "column %u out of range", type_col);
        if (m->nwide) return sk_fail(first, SK_E_STATE, "member %u has byte-string keys: no union", s);
        if (!m->d_text2 || !m->d_rank || m->no_text) return sk_fail(first, SK_E_STATE, "member %u has no text stage: no union", s);
        if (m->nrows >= (1u << SK_UNION_ROW_BITS) - 1u) return sk_fail(first, SK_E_STATE, "member %u has too many rows for the union's hit log", s);
        rows += m->nrows;
        tbases += ((uint64_t)m->text_bases + 63u) / 64u * 64u + 64u;
    }
    if (rows > 0x7FFFFFF0ull || tbases > 0x7FFFFF00ull) return sk_fail(first, SK_E_STATE, "union too large (%llu rows, %llu bases)", (unsigned long long)rows, (unsigned long long)tbases);
    SK_HIP(first, hipSetDevice(first->device));
    sk_union *u = new (std::nothrow) sk_union();
    if (!u) return SK_E_NOMEM;
    u->uc = NULL; u->n = n; u->d_ukeys = NULL; u->d_umask = NULL; u->d_members = NULL;
    u->d_tally = u->d_flag = u->d_raw = NULL; u->tally_cap = u->flag_cap = u->raw_cap = 0; u->d_cnt = NULL;
    int rc = sk_ctx_create(&u->uc, first->device);
    if (rc != SK_OK) { delete u; return rc; }
    sk_ctx *c = u->uc;
#define SK_U(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { sk_fail(first, SK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); sk_union_destroy(u); return SK_E_HIP; } } while (0)
    const uint32_t nrows = (uint32_t)rows;
    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)first->table_load_pct < rows * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    c->slots_log2 = lg;
    c->nrows = nrows; c->ncols = 1;
    SK_U(hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    SK_U(hipMalloc((void **)&u->d_ukeys, (size_t)nrows * 8));
    SK_U(hipMalloc((void **)&u->d_umask, (size_t)nrows * 8));
    uint32_t *d_canon = NULL;
    SK_U(hipMalloc((void **)&d_canon, (size_t)nrows * 4));
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, u->d_ukeys, (uint64_t)nrows, SK_EMPTY64);
    SK_U(hipMemsetAsync(u->d_umask, 0, (size_t)nrows * 8, c->stream));
    // text and rank map, member behind member
    const uint32_t text_bases = (uint32_t)tbases;
    const size_t words = (size_t)text_bases / 16 + 4, nblk = (size_t)text_bases / 64 + 2;
    SK_U(hipMalloc((void **)&c->d_text2, words * 4));
    SK_U(hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    SK_U(hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    std::vector<sk_union_member> hm(n);
    uint32_t base = 0, tbase = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        int frc = sk_diff_flush(m);                         // (the type column is read below)
        if (frc != SK_OK) { hipFree(d_canon); sk_union_destroy(u); return frc; }
        SK_U(hipStreamSynchronize(m->stream));
        const uint64_t mslots = (uint64_t)1 << m->slots_log2;
        hipLaunchKernelGGL(sk_union_insert, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)m->d_keys, mslots, base, tbase,
                           c->d_keys, (uint32_t)(slots - 1), u->d_ukeys);
        const uint32_t span = (uint32_t)(((uint64_t)m->text_bases + 63u) / 64u * 64u + 64u);
        SK_U(hipMemcpyAsync(c->d_text2 + tbase / 16u, m->d_text2, (((size_t)m->text_bases + 15) / 16) * 4, hipMemcpyDeviceToDevice, c->stream));
        const uint32_t nsrc = m->text_bases / 64u + 1u, ndst = span / 64u;
        hipLaunchKernelGGL(sk_union_rank_copy, dim3((ndst + 255) / 256), dim3(256), 0, c->stream, c->d_rank + tbase / 64u, (const sk_u4 *)m->d_rank,
                           nsrc < ndst ? nsrc : ndst, ndst, base, base + m->nrows);
        hm[s].slots = m->d_keys; hm[s].mask = (uint32_t)(mslots - 1); hm[s].inv = m->d_inv;
        base += m->nrows;
        tbase += span;
    }
    hipLaunchKernelGGL(sk_union_rank_copy, dim3(1), dim3(256), 0, c->stream, c->d_rank + tbase / 64u, (const sk_u4 *)c->d_rank, 0u, (uint32_t)(nblk - tbase / 64u), 0u, nrows);
    base = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        hipLaunchKernelGGL(sk_union_mask_a, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)m->d_keys, (uint64_t)1 << m->slots_log2, base, s,
                           (const uint32_t *)(m->d_counts + (size_t)type_col * m->nrows), informative_value,
                           (const sk_u4 *)c->d_keys, (uint32_t)(slots - 1), u->d_umask, d_canon);
        base += m->nrows;
    }
    hipLaunchKernelGGL(sk_union_mask_b, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, u->d_umask, (const uint32_t *)d_canon, nrows);
    {   // the filters, sized for all the keys (level 1 no longer fits the L2: its misses go to the Infinity Cache)
        uint64_t kib = first->grid_kib > 0 ? (uint64_t)first->grid_kib * n : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
        if (kib > (1ull << 22)) kib = 1ull << 22;
        if (kib < 4ull) kib = 4ull;
        c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
        uint32_t g2 = 12;
        while (g2 < 37 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
        c->grid2_blocks_log2 = g2 - 6u;
        const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
        SK_U(hipMalloc((void **)&c->d_grid1, b1));
        SK_U(hipMalloc((void **)&c->d_grid2, b2));
        SK_U(hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
        SK_U(hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
        bool from_text = true;                              // (every member's keys are windows of its text: one insert per base of the union's text)
        for (uint32_t s2 = 0; s2 < n; s2++) from_text = from_text && members[s2]->all_rows_in_text;
        if (from_text)
            hipLaunchKernelGGL(sk_grid_insert_text, dim3((text_bases + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->d_text2, text_bases,
                               (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        else
            hipLaunchKernelGGL(sk_grid_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)u->d_ukeys, nrows,
                               (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
    }
    SK_U(hipMalloc((void **)&u->d_members, n * sizeof(sk_union_member)));
    SK_U(hipMemcpyAsync(u->d_members, hm.data(), n * sizeof(sk_union_member), hipMemcpyHostToDevice, c->stream));
    SK_U(hipStreamSynchronize(c->stream));
    hipFree(d_canon);
    SK_U(hipGetLastError());
#undef SK_U
    c->text_bases = text_bases;
    *out = u;
    return SK_OK;
}

// grow-only device buffer of a union that must read as zero when it is handed out
static int sk_union_zeroed(sk_union *u, void **p, size_t *cap, size_t need)
{
    sk_ctx *c = u->uc;
    if (need <= *cap) return SK_OK;
    if (*p) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(*p); *p = NULL; *cap = 0; }
    const size_t want = need + need / 4 + 4096;
    SK_HIP(c, hipMalloc(p, want));
    SK_HIP(c, hipMemsetAsync(*p, 0, want, c->stream));
    *cap = want;
    return SK_OK;
}

// Start the tallies of batch `b` against every member at once; returns at once.  Results: sk_union_tally_collect.
extern "C" int sk_union_tally_launch(sk_union *u, const sk_batch *b, uint64_t hits_cap)
{
    if (!u || !b) return SK_E_ARG;
    sk_ctx *c = u->uc;
    if (b->owner->device != c->device) return sk_fail(c, SK_E_ARG, "batch lives on another device");
    if (b->nrec == 0) return sk_fail(c, SK_E_STATE, "empty batch");
    const uint64_t vrec = (uint64_t)b->nrec * u->n;
    if (vrec > 0x7FFFFFF0ull) return sk_fail(c, SK_E_ARG, "too many records x members");
    SK_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = sk_union_zeroed(u, &u->d_tally, &u->tally_cap, (size_t)vrec * 8 + 16)) != SK_OK) return rc;
    if ((rc = sk_union_zeroed(u, &u->d_flag, &u->flag_cap, (size_t)b->nrec + 16)) != SK_OK) return rc;
    if (!u->d_cnt) SK_HIP(c, hipMalloc((void **)&u->d_cnt, 32));
    if ((rc = sk_scratch(c, &u->d_raw, &u->raw_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_compact, &c->t_compact_cap, (size_t)vrec * 12 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_hits, &c->t_hits_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    // the page-locked landing area: the three counters, then room for the first SK_UNION_EAGER records and log entries, which come
    // back WITH the counters (round 4: collecting used to be three dependent trips -- wait, copy the records, copy the log, the
    // last two through pageable memory; a 32 MiB chunk against 32 strains brings ~4 K records and ~5 K entries, far below the room)
    const size_t land = 64 + (size_t)SK_UNION_EAGER * (sizeof(sk_tally_rec) + sizeof(uint2));
    if (c->h_tally_cap < land) {
        if (c->h_tally) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_tally); c->h_tally = NULL; c->h_tally_cap = 0; }
        SK_HIP(c, hipHostMalloc((void **)&c->h_tally, land, hipHostMallocDefault));
        c->h_tally_cap = land;
    }
    SK_HIP(c, hipStreamWaitEvent(c->stream, b->ready, 0));
    SK_HIP(c, hipMemsetAsync(u->d_cnt, 0, 32, c->stream));
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    sink.rec_start = (const uint32_t *)b->d_rec; sink.nrec = b->nrec; sink.tally = (uint32_t *)u->d_tally;
    sink.tile_first = (const uint32_t *)b->d_rec + b->nrec;
    sink.hits = (uint2 *)u->d_raw; sink.nhits = u->d_cnt; sink.hits_cap = hits_cap;
    sink.type = (const uint32_t *)u->d_umask; sink.ns = u->n;
    sink.infbits = (const uint32_t *)u->d_flag;                   // (sk_uflag: the records' touched marks)
    rc = sk_launch_scan(c, (const uint8_t *)b->d_stream, b->nbytes, 0, 0, &sink);
    if (rc) return rc;
    hipLaunchKernelGGL(sk_union_compact, dim3((b->nrec + 255) / 256), dim3(256), 0, c->stream, (uint2 *)u->d_tally, (uint8_t *)u->d_flag,
                       b->nrec, u->n, (uint32_t *)c->t_compact, u->d_cnt + 1);
    hipLaunchKernelGGL(sk_union_resolve, dim3(256), dim3(256), 0, c->stream, (const uint2 *)u->d_raw, (const unsigned long long *)u->d_cnt,
                       (unsigned long long)hits_cap, (uint2 *)c->t_hits, u->d_cnt + 2, (unsigned long long)hits_cap,
                       (const uint64_t *)u->d_ukeys, (const uint2 *)u->d_umask, (const sk_union_member *)u->d_members);
    SK_HIP(c, hipMemcpyAsync(c->h_tally, u->d_cnt, 24, hipMemcpyDeviceToHost, c->stream));
    {
        const size_t er = vrec < SK_UNION_EAGER ? (size_t)vrec : (size_t)SK_UNION_EAGER, eh = hits_cap < SK_UNION_EAGER ? (size_t)hits_cap : (size_t)SK_UNION_EAGER;
        if (er) SK_HIP(c, hipMemcpyAsync(c->h_tally + 64, c->t_compact, er * sizeof(sk_tally_rec), hipMemcpyDeviceToHost, c->stream));
        if (eh) SK_HIP(c, hipMemcpyAsync(c->h_tally + 64 + (size_t)SK_UNION_EAGER * sizeof(sk_tally_rec), c->t_hits, eh * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    }
    SK_HIP(c, hipGetLastError());
    c->t_inflight_nrec = (uint32_t)vrec;
    c->t_inflight_cap = hits_cap;
    return SK_OK;
}

// out[i] = {record * members + member, all hits, informative hits} for the pairs with at least one hit (unordered);
// out_hits[j] = {window-end offset, member << 27 | the member's own row}.  *out_nhits may exceed the launch's hits_cap:
// launch again with more room.
extern "C" int sk_union_tally_collect(sk_union *u, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!u || !n || !out_nhits || (cap && !out)) return SK_E_ARG;
    sk_ctx *c = u->uc;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    unsigned long long cnt[3];
    memcpy(cnt, c->h_tally, 24);
    const unsigned long long take_r = cnt[1] < cap ? cnt[1] : cap;
    {   // what came back with the counters, then (rarely) the rest
        const unsigned long long have = take_r < SK_UNION_EAGER ? take_r : SK_UNION_EAGER;
        if (have) memcpy(out, c->h_tally + 64, (size_t)have * sizeof(sk_tally_rec));
        if (take_r > have) SK_HIP(c, hipMemcpy(out + have, (const sk_tally_rec *)c->t_compact + have, (size_t)(take_r - have) * sizeof(sk_tally_rec), hipMemcpyDeviceToHost));
    }
    *n = cnt[1];
    // the raw log holds one entry per hit, the caller gets one per hit AND strain: if the raw log itself ran over, entries are
    // missing from the count below -- report at least one more than the room there was
    unsigned long long nh = cnt[2];
    if (cnt[0] > c->t_inflight_cap && nh <= c->t_inflight_cap) nh = cnt[0] > nh ? cnt[0] : c->t_inflight_cap + 1;
    const unsigned long long take = nh < c->t_inflight_cap ? nh : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    {
        const unsigned long long have = take < SK_UNION_EAGER ? take : SK_UNION_EAGER;
        if (have) memcpy(out_hits, c->h_tally + 64 + (size_t)SK_UNION_EAGER * sizeof(sk_tally_rec), (size_t)have * sizeof(uint2));
        if (take > have) SK_HIP(c, hipMemcpy(out_hits + have, (const uint2 *)c->t_hits + have, (size_t)(take - have) * sizeof(uint2), hipMemcpyDeviceToHost));
    }
    *out_nhits = nh;
    return SK_OK;
}

// wait for a launch whose results nobody will collect (the batch it reads is about to be freed)
extern "C" int sk_union_sync(sk_union *u)
{
    if (!u) return SK_E_ARG;
    sk_ctx *c = u->uc;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    return SK_OK;
}

extern "C" int sk_union_scan_timing(sk_union *u, double *total_ms, uint64_t *launches, int reset)
{
    return u ? sk_scan_timing(u->uc, total_ms, launches, reset) : SK_E_ARG;
}
extern "C" const char *sk_union_last_error(const sk_union *u) { return u ? u->uc->err : "no union"; }
extern "C" uint32_t sk_union_members(const sk_union *u) { return u ? u->n : 0u; }
extern "C" uint32_t sk_union_rows(const sk_union *u) { return u ? u->uc->nrows : 0u; }

extern "C" int sk_scan_device(sk_ctx *c, const void *dev_stream, uint64_t nbytes, uint32_t col)
{
    if (!c || (!dev_stream && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (((uintptr_t)dev_stream & 15u) != 0) return sk_fail(c, SK_E_ARG, "device stream must be 16-byte aligned");
    SK_HIP(c, hipSetDevice(c->device));
    return sk_launch_scan(c, (const uint8_t *)dev_stream, nbytes, 0, col);
}

static int sk_stage_init(sk_ctx *c)
{
    if (c->h_stage[0]) return SK_OK;
    for (int i = 0; i < SK_NSTAGE; i++) {
        SK_HIP(c, hipHostMalloc((void **)&c->h_stage[i], SK_STAGE_BYTES, hipHostMallocDefault));
        SK_HIP(c, hipMalloc((void **)&c->d_stage[i], SK_STAGE_BYTES));
        SK_HIP(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
    }
    return SK_OK;
}

extern "C" int sk_scan_stream(sk_ctx *c, const uint8_t *stream, uint64_t nbytes, uint32_t col)
{
    if (!c || (!stream && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    SK_HIP(c, hipSetDevice(c->device));
    int rc = sk_stage_init(c);
    if (rc) return rc;
    // cut into staging-sized pieces; a piece after the first re-sends the k-1 bytes before it
    uint64_t done = 0;
    while (done < nbytes) {
        const uint64_t lead = done ? SK_OVERLAP : 0;
        uint64_t take = nbytes - done;
        if (take > SK_STAGE_BYTES - lead) take = SK_STAGE_BYTES - lead;
        const int b = c->stage_next;
        c->stage_next = (b + 1) % SK_NSTAGE;
        SK_HIP(c, hipEventSynchronize(c->stage_done[b]));       // buffer free again?
        memcpy(c->h_stage[b], stream + done - lead, lead + take);
        SK_HIP(c, hipMemcpyAsync(c->d_stage[b], c->h_stage[b], lead + take, hipMemcpyHostToDevice, c->stream));
        rc = sk_launch_scan(c, c->d_stage[b], lead + take, lead, col);
        if (rc) return rc;
        SK_HIP(c, hipEventRecord(c->stage_done[b], c->stream));
        done += take;
    }
    return SK_OK;
}

// ---- zero-copy variant for callers that fill PINNED host buffers themselves ---------------------
// Page-locked buffers are kept by the context and handed out again: locking pages is slow and serialised in the driver
// (measured on the MI355X box, tools/probes/pin_probe.hip: hipHostMalloc 6.1 ms per 32 MiB plus 3.7 ms to free it, 32 buffers
// from 16 threads 189 ms -- a list scan with 16 decode threads spent 0.3 s on its buffers, every call); registering ordinary
// memory costs 2.7 ms per 32 MiB, and a buffer given back stays registered until the context goes.  May be called from
// several threads at once.
extern "C" int sk_pinned_alloc(sk_ctx *c, void **p, uint64_t nbytes)
{
    if (!c || !p) return SK_E_ARG;
    const size_t want = (size_t)((nbytes ? nbytes : 16) + 4095u) & ~(size_t)4095u;
    pthread_mutex_lock(&c->pin_mu);
    for (sk_pin &q : c->pins)
        if (!q.used && q.n == want) { q.used = true; *p = q.p; pthread_mutex_unlock(&c->pin_mu); return SK_OK; }
    pthread_mutex_unlock(&c->pin_mu);
    if (hipSetDevice(c->device) != hipSuccess) return SK_E_HIP;
    void *m = NULL;
    bool registered = true;
    if (posix_memalign(&m, 4096, want) != 0) return SK_E_NOMEM;
    // (portable: every device of the process may DMA from it -- one decoded chunk goes up to several GPUs, sk_host_sd.c)
    if (hipHostRegister(m, want, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {      // (a limit on registered memory, say): the slower way
        free(m);
        m = NULL;
        registered = false;
        (void)hipGetLastError();
        if (hipHostMalloc(&m, want, hipHostMallocPortable) != hipSuccess) return sk_fail(c, SK_E_NOMEM, "no page-locked memory (%zu bytes)", want);
    }
    pthread_mutex_lock(&c->pin_mu);
    c->pins.push_back((sk_pin){m, want, true, registered});
    pthread_mutex_unlock(&c->pin_mu);
    *p = m;
    return SK_OK;
}

extern "C" int sk_pinned_free(sk_ctx *c, void *p)
{
    if (!c) return SK_E_ARG;
    if (!p) return SK_OK;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return SK_E_HIP;   // (nothing reads it any more)
    int rc = SK_E_ARG;
    pthread_mutex_lock(&c->pin_mu);
    for (sk_pin &q : c->pins) if (q.p == p && q.used) { q.used = false; rc = SK_OK; break; }
    pthread_mutex_unlock(&c->pin_mu);
    return rc;
}

// Like sk_scan_stream, but `pinned` (from sk_pinned_alloc, at most 64 MiB - 64 bytes) is DMA-read in
// place: the caller must leave it alone until sk_ticket_wait(*ticket) returns.
extern "C" int sk_scan_pinned(sk_ctx *c, const uint8_t *pinned, uint64_t nbytes, uint32_t col, uint64_t *ticket)
{
    if (!c || (!pinned && nbytes) || !ticket) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (nbytes > SK_STAGE_BYTES - 64) return sk_fail(c, SK_E_ARG, "pinned batch larger than the staging buffer");
    SK_HIP(c, hipSetDevice(c->device));
    int rc = sk_stage_init(c);
    if (rc) return rc;
    const uint64_t t = c->tickets++;
    hipEvent_t &ev = c->copied[t & 63u];
    if (!ev) SK_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else SK_HIP(c, hipEventSynchronize(ev));                     // ring slot of ticket t-64
    const int b = c->stage_next;
    c->stage_next = (b + 1) % SK_NSTAGE;
    SK_HIP(c, hipEventSynchronize(c->stage_done[b]));
    if (nbytes) SK_HIP(c, hipMemcpyAsync(c->d_stage[b], pinned, nbytes, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipEventRecord(ev, c->stream));
    rc = sk_launch_scan(c, c->d_stage[b], nbytes, 0, col);
    if (rc) return rc;
    SK_HIP(c, hipEventRecord(c->stage_done[b], c->stream));
    *ticket = t;
    return SK_OK;
}

extern "C" int sk_ticket_wait(sk_ctx *c, uint64_t ticket)
{
    if (!c || ticket >= c->tickets) return SK_E_ARG;
    if (c->tickets - ticket > 64) return SK_OK;                  // its ring slot was recycled only after it completed
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipEventSynchronize(c->copied[ticket & 63u]));
    return SK_OK;
}

extern "C" int sk_sync(sk_ctx *c)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_fetch(sk_ctx *c, uint32_t col, uint32_t *out)
{
    if (!c || !out) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    const uint32_t *src = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows) {
        hipLaunchKernelGGL(sk_gather_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, c->d_tmp, src, c->d_perm, c->nrows);
        src = c->d_tmp;
    }
    SK_HIP(c, hipMemcpyAsync(out, src, (size_t)c->nrows * 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_set(sk_ctx *c, uint32_t col, const uint32_t *in)
{
    if (!c || !in) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    uint32_t *dst = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows) {
        SK_HIP(c, hipMemcpyAsync(c->d_tmp, in, (size_t)c->nrows * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(sk_scatter_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, dst, c->d_tmp, c->d_perm, c->nrows);
    } else {
        SK_HIP(c, hipMemcpyAsync(dst, in, (size_t)c->nrows * 4, hipMemcpyHostToDevice, c->stream));
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// counts[col][rows[i]] = value for n rows (caller's row numbers): what a column of mostly one value needs instead of 4 bytes per row
// over PCIe (strain_detect's type column: "informative" for the 1 % of rows the -a list names; src/strain_detect.c:668-726)
extern "C" int sk_counts_set_rows(sk_ctx *c, uint32_t col, const uint32_t *rows, uint32_t n, uint32_t value)
{
    if (!c || (n && !rows)) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    for (uint32_t i = 0; i < n; i++) if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "row %u out of range", rows[i]);
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    if (!n) return SK_OK;
    uint32_t *d_rows = NULL;
    SK_HIP(c, hipMalloc((void **)&d_rows, (size_t)n * 4));
    hipError_t e = hipMemcpyAsync(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sk_set_rows_u32, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_counts + (size_t)col * c->nrows, (const uint32_t *)d_rows, n,
                           (const uint32_t *)c->d_perm, value);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_rows);
    SK_HIP(c, e);
    return SK_OK;
}

extern "C" int sk_counts_zero(sk_ctx *c, uint32_t col)
{
    if (!c) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    SK_HIP(c, hipMemsetAsync(c->d_counts + (size_t)col * c->nrows, 0, (size_t)c->nrows * 4, c->stream));
    return SK_OK;
}

extern "C" void *sk_counts_device_ptr(sk_ctx *c)
{
    if (!c) return NULL;
    // whoever takes the pointer reads the block on a stream of their own: fold the pending increments in and wait
    if (hipSetDevice(c->device) != hipSuccess || sk_diff_flush(c) != SK_OK || hipStreamSynchronize(c->stream) != hipSuccess) return NULL;
    return (void *)c->d_counts;
}

// ---- for the other translation units of the library (sk_internal.h)
extern "C" int sk_fail_(sk_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}
extern "C" int sk_ctx_device_(const sk_ctx *c) { return c->device; }
extern "C" int sk_counts_rows_to_device_(sk_ctx *c, uint32_t col, uint32_t *d_out)
{
    if (!c || !d_out) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    const uint32_t *src = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows)
        hipLaunchKernelGGL(sk_gather_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, d_out, src, c->d_perm, c->nrows);
    else
        SK_HIP(c, hipMemcpyAsync(d_out, src, (size_t)c->nrows * 4, hipMemcpyDeviceToDevice, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}
extern "C" uint32_t sk_table_rows(const sk_ctx *c) { return c ? c->nrows : 0; }
extern "C" uint32_t sk_table_cols(const sk_ctx *c) { return c ? c->ncols : 0; }

extern "C" int sk_scan_timing(sk_ctx *c, double *total_ms, uint64_t *launches, int reset)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i + 1 < c->ev.size(); i += 2) {
        float ms = 0.f;
        SK_HIP(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->timed_ms += ms;
        c->timed_launches++;
        c->ev_free.push_back(c->ev[i]);
        c->ev_free.push_back(c->ev[i + 1]);
    }
    c->ev.clear();
    if (total_ms) *total_ms = c->timed_ms;
    if (launches) *launches = c->timed_launches;
    if (reset) { c->timed_ms = 0; c->timed_launches = 0; }
    return SK_OK;
}

// ---------------------------------------------------------------------------------------------
// RCCL (resolved lazily so that the library loads on hosts without it)
// ---------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } sk_nccl_id;                       // ncclUniqueId
typedef int (*sk_nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*sk_nccl_getid_fn)(sk_nccl_id *);
typedef int (*sk_nccl_initrank_fn)(void **, int, sk_nccl_id, int);
typedef int (*sk_nccl_destroy_fn)(void *);
static struct { void *lib; sk_nccl_allreduce_fn allreduce; sk_nccl_getid_fn getid; sk_nccl_initrank_fn initrank; sk_nccl_destroy_fn destroy; } g_rccl;

static int sk_rccl_load(sk_ctx *c)
{
    if (g_rccl.lib) return SK_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return sk_fail(c, SK_E_RCCL, "cannot load librccl: %s", dlerror());
    g_rccl.allreduce = (sk_nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.getid = (sk_nccl_getid_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.initrank = (sk_nccl_initrank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (sk_nccl_destroy_fn)dlsym(h, "ncclCommDestroy");
    if (!g_rccl.allreduce || !g_rccl.getid || !g_rccl.initrank || !g_rccl.destroy) return sk_fail(c, SK_E_RCCL, "RCCL symbols missing");
    g_rccl.lib = h;
    return SK_OK;
}

// One process per GPU.  The ranks first exchange, through files next to `id_file` (sk_rendezvous.h: fresh-token handshake,
// every wait bounded), whether their own set-up worked and rank 0's RCCL unique id; only if everybody is fine does anyone
// enter ncclCommInitRank, and a watchdog ends the process if that call does not return in time (a rank that died after
// the exchange would otherwise hang the rest for good).  New relative to the reference, which is single-process.
struct sk_watchdog { pthread_mutex_t mu; pthread_cond_t cv; int done; int timeout_s; int rank; };
static void *sk_watchdog_main(void *arg)
{
    sk_watchdog *w = (sk_watchdog *)arg;
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += w->timeout_s;
    pthread_mutex_lock(&w->mu);
    while (!w->done) {
        if (pthread_cond_timedwait(&w->cv, &w->mu, &until) == ETIMEDOUT && !w->done) {
            fprintf(stderr, "sk_comm_init: ncclCommInitRank did not return within %d s on rank %d (did a rank die after the rendezvous?) -- giving up\n",
                    w->timeout_s, w->rank);
            fflush(stderr);
            _exit(3);                                  // (never an exec: this process has touched the GPU)
        }
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

extern "C" int sk_comm_init_ex(sk_ctx *c, int rank, int world, const char *id_file, int timeout_s, int setup_failed)
{
    if (world < 1 || world > SKR_MAX_WORLD || rank < 0 || rank >= world || !id_file || (!c && !setup_failed)) return SK_E_ARG;
    if (timeout_s < 1) timeout_s = 1;
    sk_nccl_id id;
    memset(&id, 0, sizeof id);
    int failed = setup_failed != 0;
    if (!failed && sk_rccl_load(c) != SK_OK) failed = 1;
    if (!failed && hipSetDevice(c->device) != hipSuccess) failed = 1;
    if (!failed && rank == 0 && g_rccl.getid(&id) != 0) { sk_fail(c, SK_E_RCCL, "ncclGetUniqueId failed"); failed = 1; }
    static_assert(sizeof id == SKR_PAYLOAD, "RCCL unique id size");
    const int x = skr_exchange(rank, world, id_file, failed, (unsigned char *)&id, (double)timeout_s);
    if (rank == 0) {                                   // (the others remove their own hello files)
        // the board stays until the ranks have read it; it carries this launch's tokens, so a later launch ignores it
    }
    if (x == SKR_ABORT) return sk_fail(c, SK_E_RCCL, failed ? "set-up failed on this rank; the other ranks were told to leave"
                                                            : "another rank reported a failed set-up: leaving before the collective");
    if (x == SKR_TIMEOUT) return sk_fail(c, SK_E_RCCL, "rendezvous through %s timed out after %d s (rank %d of %d)", id_file, timeout_s, rank, world);
    if (x != SKR_OK) return sk_fail(c, SK_E_RCCL, "rendezvous through %s failed (cannot write there?)", id_file);
    sk_watchdog w;
    pthread_mutex_init(&w.mu, NULL); pthread_cond_init(&w.cv, NULL);
    w.done = 0; w.timeout_s = timeout_s; w.rank = rank;
    pthread_t th;
    const bool watched = pthread_create(&th, NULL, sk_watchdog_main, &w) == 0;
    void *comm = NULL;
    const int nrc = g_rccl.initrank(&comm, world, id, rank);
    if (watched) {
        pthread_mutex_lock(&w.mu); w.done = 1; pthread_cond_signal(&w.cv); pthread_mutex_unlock(&w.mu);
        pthread_join(th, NULL);
    }
    pthread_mutex_destroy(&w.mu); pthread_cond_destroy(&w.cv);
    if (nrc != 0) return sk_fail(c, SK_E_RCCL, "ncclCommInitRank failed (rank %d of %d)", rank, world);
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return SK_OK;
}

extern "C" int sk_comm_init(sk_ctx *c, int rank, int world, const char *id_file, int timeout_s)
{
    if (!c) return SK_E_ARG;
    return sk_comm_init_ex(c, rank, world, id_file, timeout_s, 0);
}

extern "C" void sk_comm_destroy(sk_ctx *c)
{
    if (c && c->comm && g_rccl.destroy) { g_rccl.destroy(c->comm); c->comm = NULL; }
}

// Sum a small host value over all ranks (agreement on "did anyone fail" before the big collective).
extern "C" int sk_comm_sum_u32(sk_ctx *c, uint32_t value, uint32_t *sum)
{
    if (!c || !sum) return SK_E_ARG;
    if (!c->comm) { *sum = value; return SK_OK; }
    SK_HIP(c, hipSetDevice(c->device));
    uint32_t *d = c->d_flags + 12;                     // spare words of the flag block
    SK_HIP(c, hipMemcpyAsync(d, &value, 4, hipMemcpyHostToDevice, c->stream));
    const int ncclUint32 = 3, ncclSum = 0;
    if (g_rccl.allreduce(d, d, 1, ncclUint32, ncclSum, c->comm, c->stream) != 0) return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipMemcpyAsync(sum, d, 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// Element-wise maximum of up to 8 host words over all ranks, in place: the one small collective the list walk is built on
// (plan hash + local failure flags before a list is scanned, what went wrong where after it: every rank issues the same
// sequence of these whatever happens to it locally).  No communicator: the values stay as they are.
extern "C" int sk_comm_max_u64(sk_ctx *c, uint64_t *vals, uint32_t n)
{
    if (!vals || n == 0 || n > 8) return SK_E_ARG;
    if (!c || !c->comm) return SK_OK;
    SK_HIP(c, hipSetDevice(c->device));
    uint64_t *d = (uint64_t *)(c->d_flags + 16);      // scratch words of the flag block (64 bytes in: 8-byte aligned)
    SK_HIP(c, hipMemcpyAsync(d, vals, 8u * n, hipMemcpyHostToDevice, c->stream));
    const int ncclUint64 = 5, ncclMax = 2;            // rccl.h: ncclDataType_t / ncclRedOp_t
    if (g_rccl.allreduce(d, d, n, ncclUint64, ncclMax, c->comm, c->stream) != 0) return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipMemcpyAsync(vals, d, 8u * n, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// ranks of the context's communicator (0: none) -- the list walk coordinates its fall-backs only when every rank of the run is in it
extern "C" int sk_comm_world(const sk_ctx *c) { return c && c->comm ? c->comm_world : 0; }

// Do all ranks hold the same 64-bit value (a hash of the work plan, before anyone scans)?  One max all-reduce over
// {v, ~v}: the values agree iff max(v) == min(v) == ~max(~v).  No communicator: a world of one agrees with itself.
extern "C" int sk_comm_agree_u64(sk_ctx *c, uint64_t value, int *agree)
{
    if (!agree) return SK_E_ARG;
    *agree = 1;
    uint64_t h[2] = {value, ~value};
    const int rc = sk_comm_max_u64(c, h, 2);
    if (rc != SK_OK) return rc;
    *agree = h[0] == value && h[1] == ~value;         // (max v == v and min v == v on this rank <=> on every rank)
    return SK_OK;
}

// In-library sum all-reduce of the whole counter block; rccl_comm == NULL uses sk_comm_init's.
extern "C" int sk_counts_allreduce(sk_ctx *c, void *rccl_comm)
{
    if (!c) return SK_E_ARG;
    if (!c->d_counts) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (!rccl_comm) rccl_comm = c->comm;
    if (!rccl_comm) return sk_fail(c, SK_E_STATE, "no communicator");
    int rc = sk_rccl_load(c);
    if (rc) return rc;
    SK_HIP(c, hipSetDevice(c->device));
    if ((rc = sk_diff_flush(c)) != SK_OK) return rc;
    const int ncclUint32 = 3, ncclSum = 0;            // rccl.h: ncclDataType_t / ncclRedOp_t
    if (g_rccl.allreduce(c->d_counts, c->d_counts, (size_t)c->nrows * c->ncols, ncclUint32, ncclSum, rccl_comm, c->stream) != 0)
        return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_dev_alloc(sk_ctx *c, void **dev, uint64_t nbytes)
{
    if (!c || !dev) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    if (c->dev_uncached) SK_HIP(c, hipExtMallocWithFlags(dev, nbytes ? nbytes : 16, hipDeviceMallocUncached));
    else SK_HIP(c, hipMalloc(dev, nbytes ? nbytes : 16));
    return SK_OK;
}

extern "C" int sk_dev_free(sk_ctx *c, void *dev)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipFree(dev));
    return SK_OK;
}

extern "C" int sk_dev_upload(sk_ctx *c, void *dev, const void *host, uint64_t nbytes)
{
    if (!c || !dev || !host) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_dev_download(sk_ctx *c, void *host, const void *dev, uint64_t nbytes)
{
    if (!c || !dev || !host) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

#if SK_PHASE_CLOCK
// experiment builds only: the eight phase sums (wave-cycles of s_memtime) since the last call, then zeroed
extern "C" int sk_debug_phase_clock(sk_ctx *c, unsigned long long out[8])
{
    if (!c || !out) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long *d = (unsigned long long *)(c->d_oddlist + (c->odd_cap ? c->odd_cap : SK_ODDCAP) - 1024u), h[512];
    SK_HIP(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    SK_HIP(c, hipMemset(d, 0, sizeof h));
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int s = 0; s < 64; s++) out[k] += h[s * 8 + k]; }
    return SK_OK;
}
#endif
