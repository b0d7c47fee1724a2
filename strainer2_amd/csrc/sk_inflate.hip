// sk_inflate.hip -- gzip/DEFLATE on the device (EXPERIMENTAL, off unless SK_GPU_INFLATE=1; internal to the library).
//
// The host's inflate is what bounds every .gz input (0.97 GB/s of text per thread, 9.5 GB/s on the box's 16 CPUs: DESIGN.md section 5).
// DEFLATE has no entry points, so the scheme is the speculative one of sk_gzpar.h, moved to the device (prototype and measurements:
// tools/probes/gpu_inflate_spec.hip, profiles/r02_gpu_inflate_probe.txt):
//   find_candidates   a wave per 16 KiB segment runs every bit position through the cheap header checks (64 x 4 at a time)
//   validate          a lane per surviving candidate: the whole dynamic header through every check zlib makes
//   spec_decode       a lane per segment: tables of its first standing candidate, then block after block into 16-bit symbols -- a byte,
//                     or 256 + p for "the byte at place p of the 32 KiB before this segment" -- until a block ends at or behind the next
//                     segment's first bit
//   extend            lanes whose block ended in a segment without a block start of its own go on to the next segment that has one
//   (host)            the chain is CHECKED: the first segment starts at the member's first block, every other on the bit where the one
//                     before it ended, the last ends in a final block, the lengths add up to the trailer's ISIZE.  Then, by induction,
//                     the symbols are the serial decoder's.  Anything else -- a stored or fixed block, a second member, a wrong guess, an
//                     output cap -- and sk_inflate_gz returns SK_E_UNSUPPORTED: the caller decodes the file on the host, as before.
//   tails_compose / tails_chain / resolve_translate   placeholders -> bytes: "the last 32 KiB after a segment in terms of the 32 KiB
//                     before it" composes associatively, so the chain is walked in groups in parallel, then the groups, then the
//                     groups again writing bytes
// The text is copied into the caller's (page-locked) buffer; its CRC-32 is the caller's to check.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "../../include/strainer_kmer.h"
#include "sk_internal.h"

#define SKI_FIRST_BLOCK_MAX 300000u       // symbols a candidate's first block may decode to while it is still only a candidate
#define SKI_SEG_BYTES (16u << 10)
#define SKI_MAX_GZ (256ull << 20)          // bigger files: the host (sk_gzpar.h inflates one big file on all threads)

#define LB 11
#define DB 8
#define NLIT (2048 + 286 * 16)
#define NDIST (256 + 30 * 128)
#define WINDOW 32768u
enum { K_LIT = 0, K_LIT2 = 1, K_LEN = 2, K_EOB = 3, K_SUB = 4, K_BAD = 5, K_DIST = 6 };
#define ENTRY(val, extra, kind, nbits) (((uint32_t)(val) << 16) | ((uint32_t)(extra) << 8) | ((uint32_t)(kind) << 4) | (uint32_t)(nbits))

struct seg_out { uint64_t start_bit, end_bit; uint32_t n, flags, blocks, tried; };   // flags: 1 ok, 2 final block seen, 4 hit a stored/fixed block, 8 output cap, 16 nothing found

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t  c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t  c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t  c_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// bit reader over the file as 32-bit words (zeros past its end)
struct bits {
    const uint32_t *w; uint64_t nwords, wp; uint64_t buf; uint32_t cnt;
    __device__ void seek(uint64_t bit) { wp = bit >> 5; buf = (uint64_t)(wp < nwords ? w[wp] : 0u) >> (bit & 31u); cnt = 32u - (uint32_t)(bit & 31u); wp++; }
    __device__ void fill() { if (cnt <= 32u) { buf |= (uint64_t)(wp < nwords ? w[wp] : 0u) << cnt; wp++; cnt += 32u; } }
    __device__ uint32_t peek(uint32_t n) const { return (uint32_t)(buf & (((uint64_t)1 << n) - 1u)); }
    __device__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ uint64_t pos() const { return wp * 32u - cnt; }
};

__device__ uint32_t rev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }

__device__ uint32_t sym_entry(int is_dist, uint32_t sym, uint32_t nbits)
{
    if (is_dist) return sym < 30 ? ENTRY(c_dist_base[sym], c_dist_extra[sym], K_DIST, nbits) : ENTRY(0, 0, K_BAD, nbits);
    if (sym < 256) return ENTRY(sym, 0, K_LIT, nbits);
    if (sym == 256) return ENTRY(0, 0, K_EOB, nbits);
    if (sym < 286) return ENTRY(c_len_base[sym - 257], c_len_extra[sym - 257], K_LEN, nbits);
    return ENTRY(0, 0, K_BAD, nbits);
}

// canonical Huffman decode table (primary of tbits + subtables), as skz_build of sk_gzfast.h without the literal pairs
__device__ int build(uint32_t *table, uint32_t tbits, uint32_t cap, const uint8_t *lens, uint32_t nsym, int is_dist)
{
    uint32_t count[16], next[16], next2[16];
    for (int i = 0; i < 16; i++) count[i] = 0;
    for (uint32_t s = 0; s < nsym; s++) count[lens[s]]++;
    count[0] = 0;
    uint32_t left = 1, code = 0;
    for (uint32_t len = 1; len <= 15; len++) { left <<= 1; if (count[len] > left) return -1; left -= count[len]; }
    for (uint32_t len = 1; len <= 15; len++) { code = (code + count[len - 1]) << 1; next[len] = next2[len] = code; }
    const uint32_t prim = 1u << tbits;
    for (uint32_t i = 0; i < prim; i++) table[i] = ENTRY(0, 0, K_BAD, 1);
    // pass 1: short codes; the longest code through every primary slot is kept in the slot itself (kind K_BAD, value = depth)
    for (uint32_t s = 0; s < nsym; s++) {
        const uint32_t len = lens[s];
        if (!len) continue;
        const uint32_t r = rev(next[len]++, len);
        if (len <= tbits) { const uint32_t e = sym_entry(is_dist, s, len); for (uint32_t i = r; i < prim; i += 1u << len) table[i] = e; }
        else { const uint32_t p = r & (prim - 1u); const uint32_t d = table[p] >> 16; if (len > d) table[p] = ENTRY(len, 0, K_BAD, 1); }
    }
    uint32_t used = prim;
    for (uint32_t i = 0; i < prim; i++) {
        const uint32_t d = table[i] >> 16;
        if (((table[i] >> 4) & 15u) == K_BAD && d > tbits) {
            const uint32_t sbits = d - tbits;
            if (used + (1u << sbits) > cap || used > 0xFFFFu) return -1;
            table[i] = ENTRY(used, sbits, K_SUB, tbits);
            for (uint32_t j = 0; j < (1u << sbits); j++) table[used + j] = ENTRY(0, 0, K_BAD, 1);
            used += 1u << sbits;
        }
    }
    for (uint32_t s = 0; s < nsym; s++) {
        const uint32_t len = lens[s];
        if (len <= tbits) continue;
        const uint32_t r = rev(next2[len]++, len), p = r & (prim - 1u);
        const uint32_t start = table[p] >> 16, sbits = (table[p] >> 8) & 255u, e = sym_entry(is_dist, s, len - tbits);
        for (uint32_t i = r >> tbits; i < (1u << sbits); i += 1u << (len - tbits)) table[start + i] = e;
    }
    // literal pairs, as the host's tables have them: where the bits of a primary index spell one literal and then, completely, a
    // second one, the entry yields both.  In place, from the high indices down?  No: entry i looks at entry i >> l1 (a LOWER index
    // unless l1 = 0), which must still be the single-literal entry -- so a first pass marks (kind K_LIT2 needs both literals in the
    // value), reading only K_LIT entries' original values kept in the low byte of the value field of what it reads.
    if (!is_dist) {
        for (uint32_t i = prim; i-- > 0;) {                                    // descending: entry i >> l1 <= i is rewritten only after i was
            const uint32_t e1 = table[i], l1 = e1 & 15u;
            if (((e1 >> 4) & 15u) != K_LIT || l1 >= tbits) continue;
            const uint32_t e2 = table[i >> l1];
            const uint32_t k2 = (e2 >> 4) & 15u;
            // (e2 may already be a pair when i >> l1 == i, i.e. i = 0 and... l1 > 0 makes i >> l1 < i except for i = 0: entry 0 is
            // handled last and reads itself before it is rewritten)
            if (k2 != K_LIT || (e2 & 15u) > tbits - l1) continue;
            table[i] = ENTRY(((e1 >> 16) & 255u) | (((e2 >> 16) & 255u) << 8), 0, K_LIT2, l1 + (e2 & 15u));
        }
    }
    return 0;
}

__device__ bool complete(const uint8_t *lens, uint32_t n) { uint32_t sum = 0; for (uint32_t i = 0; i < n; i++) if (lens[i]) sum += 1u << (15 - lens[i]); return sum == (1u << 15); }

// the dynamic header behind BFINAL/BTYPE: code lengths, both tables; as skz_read_dynamic(strict) of sk_gzfast.h
__device__ int read_lens(bits &b, uint8_t *lens, uint32_t &hlit_out, uint32_t &hdist_out);
__device__ int read_dynamic(bits &b, uint32_t *lt, uint32_t *dt, uint8_t *lens)
{
    uint32_t hlit, hdist;
    if (read_lens(b, lens, hlit, hdist)) return -1;
    if (build(lt, LB, NLIT, lens, hlit, 0)) return -1;
    if (build(dt, DB, NDIST, lens + hlit, hdist, 1)) return -1;
    return 0;
}
// the header's code lengths and every check on them (no tables yet)
__device__ int read_lens(bits &b, uint8_t *lens, uint32_t &hlit_out, uint32_t &hdist_out)
{
    uint8_t cl[19];
    uint16_t cltab[128];
    b.fill();
    const uint32_t hlit = b.peek(5) + 257; b.drop(5);
    const uint32_t hdist = b.peek(5) + 1; b.drop(5);
    const uint32_t hclen = b.peek(4) + 4; b.drop(4);
    if (hlit > 286 || hdist > 30) return -1;
    for (int i = 0; i < 19; i++) cl[i] = 0;
    for (uint32_t i = 0; i < hclen; i++) { b.fill(); cl[c_order[i]] = (uint8_t)b.peek(3); b.drop(3); }
    {
        uint32_t count[8], next[8], code = 0, left = 1;
        for (int i = 0; i < 8; i++) count[i] = 0;
        for (int s = 0; s < 19; s++) count[cl[s]]++;
        count[0] = 0;
        for (uint32_t len = 1; len <= 7; len++) { left <<= 1; if (count[len] > left) return -1; left -= count[len]; }
        for (uint32_t len = 1; len <= 7; len++) { code = (code + count[len - 1]) << 1; next[len] = code; }
        for (int i = 0; i < 128; i++) cltab[i] = 0xFFFFu;
        for (uint32_t s = 0; s < 19; s++) {
            const uint32_t len = cl[s];
            if (!len) continue;
            const uint32_t r = rev(next[len]++, len);
            for (uint32_t j = r; j < 128; j += 1u << len) cltab[j] = (uint16_t)((s << 8) | len);
        }
    }
    uint32_t n = 0;
    while (n < hlit + hdist) {
        b.fill();
        const uint32_t e = cltab[b.peek(7)];
        if (e == 0xFFFFu) return -1;
        b.drop(e & 255u);
        const uint32_t sym = e >> 8;
        uint32_t rep, val;
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        if (sym == 16) { if (n == 0) return -1; val = lens[n - 1]; rep = 3 + b.peek(2); b.drop(2); }
        else if (sym == 17) { val = 0; rep = 3 + b.peek(3); b.drop(3); }
        else { val = 0; rep = 11 + b.peek(7); b.drop(7); }
        if (n + rep > hlit + hdist) return -1;
        while (rep--) lens[n++] = (uint8_t)val;
    }
    if (b.pos() > b.nwords * 32u || lens[256] == 0) return -1;
    uint32_t nd = 0;
    for (uint32_t i = 0; i < hdist; i++) nd += lens[hlit + i] != 0;
    if (!complete(lens, hlit)) return -1;
    if (nd > 1 && !complete(lens + hlit, hdist)) return -1;
    hlit_out = hlit; hdist_out = hdist;
    return 0;
}

// one Huffman block into 16-bit symbols; 0 = ended on its end-of-block code, 1 = output cap, -1 = not a block
__device__ int block16(bits &b, const uint32_t *lt, const uint32_t *dt, uint16_t *o, uint32_t &n, uint32_t cap)
{
    for (;;) {
        if (n + 260u > cap) return 1;
        b.fill();
        uint32_t e = lt[b.peek(LB)];
        uint32_t kind = (e >> 4) & 15u;
        if (kind <= K_LIT2) {
            b.drop(e & 15u);
            o[n] = (uint16_t)((e >> 16) & 255u);
            o[n + 1] = (uint16_t)(e >> 24);                                    // (the second of a pair; overwritten by the next symbol otherwise)
            n += 1u + (kind == K_LIT2);
            continue;
        }
        if (kind == K_SUB) {
            b.drop(LB);
            e = lt[(e >> 16) + b.peek((e >> 8) & 255u)];
            kind = (e >> 4) & 15u;
            if (kind == K_LIT) { b.drop(e & 15u); o[n++] = (uint16_t)(e >> 16); continue; }
        }
        b.drop(e & 15u);
        if (kind == K_LEN) {
            const uint32_t xb = (e >> 8) & 255u;
            const uint32_t len = (e >> 16) + b.peek(xb);
            b.drop(xb);
            b.fill();
            uint32_t d = dt[b.peek(DB)];
            if (((d >> 4) & 15u) == K_SUB) { b.drop(DB); d = dt[(d >> 16) + b.peek((d >> 8) & 255u)]; }
            b.drop(d & 15u);
            if (((d >> 4) & 15u) != K_DIST) return -1;
            const uint32_t db = (d >> 8) & 255u;
            const uint32_t dist = (d >> 16) + b.peek(db);
            b.drop(db);
            // place n of the output = place WINDOW + n of (window, output); a source place below WINDOW is a placeholder
            for (uint32_t k = 0; k < len; k++) {
                const uint32_t src = WINDOW + n + k - dist;                  // (dist <= 32768 <= WINDOW + n: never negative)
                o[n + k] = src < WINDOW ? (uint16_t)(256u + src) : o[src - WINDOW];
            }
            n += len;
            continue;
        }
        if (kind == K_EOB) return b.pos() > b.nwords * 32u ? -1 : 0;
        return -1;
    }
}

#define CANDMAX 1024u
// every bit position of segment k (k >= 1) through the cheap checks, 64 at a time; survivors in ascending order in cand[k][..]
__global__ void find_candidates(const uint32_t *__restrict__ comp32, uint64_t nwords, uint64_t total_bits, uint32_t seg_bytes, uint64_t base_byte,
                                uint32_t nseg, uint32_t *cand, uint32_t *ncand)
{
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (k >= nseg || k == 0) return;
    const uint64_t lo = (base_byte + (uint64_t)k * seg_bytes) * 8u;
    uint64_t hi = (base_byte + (uint64_t)(k + 1) * seg_bytes) * 8u;
    if (hi > total_bits) hi = total_bits;
    uint32_t n = 0;
    for (uint64_t base = lo; base < hi; base += 256u) {                        // (wave-uniform) four bit positions per lane out of one set of loads
        const uint64_t bit0 = base + 4u * lane;
        const uint64_t wp = bit0 >> 5;
        const uint32_t sh = (uint32_t)(bit0 & 31u);
        const unsigned __int128 win = ((unsigned __int128)(wp + 3 < nwords ? comp32[wp + 3] : 0u) << 96) | ((unsigned __int128)(wp + 2 < nwords ? comp32[wp + 2] : 0u) << 64) |
                                      ((unsigned __int128)(wp + 1 < nwords ? comp32[wp + 1] : 0u) << 32) | (unsigned __int128)(wp < nwords ? comp32[wp] : 0u);
        bool ok[4];
        unsigned long long m[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const unsigned __int128 x = win >> (sh + j);                       // >= 94 bits from position bit0 + j; 74 are looked at
            const uint64_t a = (uint64_t)x, b2 = (uint64_t)(x >> 64);
            const uint32_t w = (uint32_t)a & 0x1FFFFu;
            ok[j] = bit0 + j < hi && (w & 7u) == 4u && ((w >> 3) & 31u) <= 29u && ((w >> 8) & 31u) <= 29u;
            if (ok[j]) {
                const uint32_t hclen = ((w >> 13) & 15u) + 4u;
                uint32_t sum = 0;
                for (uint32_t i = 0; i < hclen; i++) {
                    const uint32_t at = 17u + 3u * i;
                    const uint32_t l = (uint32_t)(at + 3u <= 64u ? (a >> at) : at >= 64u ? (b2 >> (at - 64u)) : ((a >> at) | (b2 << (64u - at)))) & 7u;
                    if (l) sum += 128u >> l;
                }
                ok[j] = sum == 128u;
            }
            m[j] = __ballot(ok[j]);
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t at = n + (uint32_t)(__popcll(m[0] & below) + __popcll(m[1] & below) + __popcll(m[2] & below) + __popcll(m[3] & below));
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            if (ok[j]) { if (at < CANDMAX) cand[(size_t)k * CANDMAX + at] = (uint32_t)(bit0 + j - lo); at++; }
        n += (uint32_t)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
    }
    if (lane == 0) ncand[k] = n < CANDMAX ? n : CANDMAX;
}

// a lane per CANDIDATE: the whole header through every check; the ones that fail are struck from the list (top bit).  Done apart from
// the decode so that a wave's lanes are all in the same kind of work at the same time: with the checks inside the decode lane's loop,
// every lane of a wave met its true start in another iteration and the 64 first-block decodes of a wave ran one after the other.
__global__ void validate(const uint32_t *__restrict__ comp32, uint64_t nwords, uint32_t seg_bytes, uint64_t base_byte, uint32_t nseg, uint32_t *cand, const uint32_t *__restrict__ ncand)
{
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;     // a wave per segment, its lanes over the list
    if (k >= nseg || k == 0) return;
    const uint32_t nc = ncand[k];
    for (uint32_t c = lane; c < nc; c += 64u) {
        uint8_t lens[320];
        uint32_t hlit, hdist;
        bits b; b.w = comp32; b.nwords = nwords;
        b.seek((base_byte + (uint64_t)k * seg_bytes) * 8u + cand[(size_t)k * CANDMAX + c] + 3u);
        if (read_lens(b, lens, hlit, hdist)) cand[(size_t)k * CANDMAX + c] |= 0x80000000u;
    }
}

// lane per segment: its first candidate whose header stands up to every check and whose block decodes; then block after block until
// one ends at or behind the next segment's first bit (or is final)
__global__ void spec_decode(const uint32_t *__restrict__ comp32, uint64_t nwords, uint64_t first_bit, uint64_t total_bits, uint32_t seg_bytes, uint64_t base_byte,
                            uint32_t nseg, const uint32_t *__restrict__ cand, const uint32_t *__restrict__ ncand, uint32_t *tabs, uint16_t *sym, uint32_t cap, seg_out *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nseg) return;
    uint32_t *lt = tabs + (size_t)k * (NLIT + NDIST), *dt = lt + NLIT;
    uint16_t *o = sym + (size_t)k * cap;
    uint8_t lens[320];
    const uint64_t lo = k == 0 ? first_bit : (base_byte + (uint64_t)k * seg_bytes) * 8u;
    const uint64_t stop = k + 1 < nseg ? (base_byte + (uint64_t)(k + 1) * seg_bytes) * 8u : 0;     // 0: to the final block
    bits b; b.w = comp32; b.nwords = nwords;
    seg_out r; r.start_bit = r.end_bit = 0; r.n = 0; r.flags = 0; r.blocks = 0; r.tried = 0;
    uint32_t n = 0;
    bool have = false;
    const uint32_t nc = k == 0 ? 1u : ncand[k];
    uint32_t c = 0;
    for (;;) {
        // (the cheap walk to the next candidate that validate left standing is kept apart from the decode: all lanes of a wave
        // then decode their first block in the same pass of this loop, whatever place their candidate has in its list)
        while (k && c < nc && (cand[(size_t)k * CANDMAX + c] & 0x80000000u)) c++;
        if (c >= nc) break;
        const uint64_t bit = k == 0 ? first_bit : lo + cand[(size_t)k * CANDMAX + c];
        b.seek(bit);
        b.fill();
        if (k == 0 && (b.peek(3) & 6u) != 4u) { r.flags = 4; break; }        // the first block: any BFINAL, must be dynamic for this probe
        r.tried++;
        const uint32_t final = b.peek(1);
        b.drop(3);
        if (read_dynamic(b, lt, dt, lens) == 0) {
            n = 0;
            // (a header that only looks like one decodes noise: give it the room of a big block, not the segment's whole buffer)
            const int rc = block16(b, lt, dt, o, n, k == 0 ? cap : (cap < SKI_FIRST_BLOCK_MAX ? cap : SKI_FIRST_BLOCK_MAX));
            if (rc == 0) { have = true; r.start_bit = bit; r.blocks = 1; if (final) r.flags |= 2; break; }
        }
        c++;
    }
    if (!have) { if (!r.flags) r.flags = 16; out[k] = r; return; }
    while (!(r.flags & 2)) {
        const uint64_t at = b.pos();
        if (stop && at >= stop) break;
        b.fill();
        const uint32_t final = b.peek(1), type = (b.peek(3) >> 1);
        b.drop(3);
        if (type != 2) { r.flags |= 4; break; }
        if (read_dynamic(b, lt, dt, lens)) { r.flags |= 32; break; }
        const int rc = block16(b, lt, dt, o, n, cap);
        if (rc == 1) { r.flags |= 8; break; }
        if (rc) { r.flags |= 32; break; }
        r.blocks++;
        if (final) r.flags |= 2;
    }
    r.end_bit = b.pos();
    r.n = n;
    if (!(r.flags & (4 | 8 | 32))) r.flags |= 1;
    out[k] = r;
}

// a lane whose last block ended in (or behind) a segment that has no block start of its own goes on from there until it meets the next
// segment that has one (or the final block)
__global__ void extend(const uint32_t *__restrict__ comp32, uint64_t nwords, uint32_t nseg, uint32_t *tabs, uint16_t *sym, uint32_t cap, seg_out *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nseg) return;
    seg_out r = out[k];
    if (!(r.flags & 1) || (r.flags & 2)) return;
    uint32_t j = k + 1;
    while (j < nseg && !(out[j].flags & 1)) j++;                               // the next segment with a start of its own (read-only here: flags & 1 of others never change)
    const uint64_t target = j < nseg ? out[j].start_bit : ~0ull;
    if (r.end_bit >= target) return;
    uint32_t *lt = tabs + (size_t)k * (NLIT + NDIST), *dt = lt + NLIT;
    uint16_t *o = sym + (size_t)k * cap;
    uint8_t lens[320];
    bits b; b.w = comp32; b.nwords = nwords;
    b.seek(r.end_bit);
    uint32_t n = r.n;
    while (!(r.flags & 2) && b.pos() < target) {
        b.fill();
        const uint32_t final = b.peek(1), type = (b.peek(3) >> 1);
        b.drop(3);
        if (type != 2) { r.flags = (r.flags & ~1u) | 4; break; }
        if (read_dynamic(b, lt, dt, lens)) { r.flags = (r.flags & ~1u) | 32; break; }
        const int rc = block16(b, lt, dt, o, n, cap);
        if (rc) { r.flags = (r.flags & ~1u) | (rc == 1 ? 8 : 32); break; }
        r.blocks++;
        if (final) r.flags |= 2;
    }
    r.end_bit = b.pos();
    r.n = n;
    out[k].end_bit = r.end_bit; out[k].n = r.n; out[k].blocks = r.blocks;
    out[k].flags = r.flags | 64u;                                               // 64: extended
}


// ---- placeholders -> bytes ------------------------------------------------------------------------------------------------------------
// chain[i] = segment index of the i-th segment of the chain, len/off = its symbols and where its bytes go.  T_i[j], j in 0..32767: the symbol at
// place j of the last 32 KiB of (window before segment i, output of segment i), places of the window before as 256 + place.
#define GROUP 256u
struct chain_seg { uint32_t seg, len; uint64_t off; };
__device__ __forceinline__ uint32_t tail_sym(const uint16_t *o, uint32_t len, uint32_t j)
{
    if (len >= WINDOW) return o[len - WINDOW + j];
    return j < WINDOW - len ? 256u + (j + len) : o[j - (WINDOW - len)];
}
// level 1: per group, the map "window after the group's last segment in terms of the window before its first"
__global__ void tails_compose(const chain_seg *__restrict__ ch, uint32_t nch, const uint16_t *__restrict__ sym, uint32_t cap, uint16_t *gmap)
{
    extern __shared__ uint16_t lds[];
    uint16_t *M = lds, *N = lds + WINDOW;
    const uint32_t g = blockIdx.x, first = g * GROUP, last = first + GROUP < nch ? first + GROUP : nch;
    for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) M[j] = (uint16_t)(256u + j);
    __syncthreads();
    for (uint32_t i = first; i < last; i++) {
        const uint16_t *o = sym + (size_t)ch[i].seg * cap;
        const uint32_t len = ch[i].len;
        for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) { const uint32_t t = tail_sym(o, len, j); N[j] = t < 256u ? (uint16_t)t : M[t - 256u]; }
        __syncthreads();
        uint16_t *x = M; M = N; N = x;
    }
    for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) gmap[(size_t)g * WINDOW + j] = M[j];
}
// level 2: the window before every group (bytes); the window before the first segment is empty (a valid stream never looks there)
__global__ void tails_chain(const uint16_t *__restrict__ gmap, uint32_t ngroups, uint8_t *gwin)
{
    extern __shared__ uint16_t lds[];
    uint8_t *W = (uint8_t *)lds, *V = W + WINDOW;
    for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) W[j] = 0;
    __syncthreads();
    for (uint32_t g = 0; g < ngroups; g++) {
        for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) {
            gwin[(size_t)g * WINDOW + j] = W[j];
            const uint32_t t = gmap[(size_t)g * WINDOW + j];
            V[j] = t < 256u ? (uint8_t)t : W[t - 256u];
        }
        __syncthreads();
        uint8_t *x = W; W = V; V = x;
    }
}
// level 3: per group from its window: every segment's symbols to bytes at their place in the text, then the window moves on
__global__ void resolve_translate(const chain_seg *__restrict__ ch, uint32_t nch, const uint16_t *__restrict__ sym, uint32_t cap, const uint8_t *__restrict__ gwin, uint8_t *text)
{
    extern __shared__ uint16_t lds[];
    uint8_t *W = (uint8_t *)lds, *V = W + WINDOW;
    const uint32_t g = blockIdx.x, first = g * GROUP, last = first + GROUP < nch ? first + GROUP : nch;
    for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) W[j] = gwin[(size_t)g * WINDOW + j];
    __syncthreads();
    for (uint32_t i = first; i < last; i++) {
        const uint16_t *o = sym + (size_t)ch[i].seg * cap;
        const uint32_t len = ch[i].len;
        uint8_t *dst = text + ch[i].off;
        for (uint32_t p = threadIdx.x; p < len; p += blockDim.x) { const uint32_t t = o[p]; dst[p] = t < 256u ? (uint8_t)t : W[t - 256u]; }
        for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) { const uint32_t t = tail_sym(o, len, j); V[j] = t < 256u ? (uint8_t)t : W[t - 256u]; }
        __syncthreads();
        uint8_t *x = W; W = V; V = x;
    }
}


struct sk_inflater {
    int device;
    hipStream_t stream;
    hipEvent_t done;                     // blocking waits (hipEventBlockingSync): a thread that feeds the device sleeps meanwhile
    void *buf[10]; size_t cap[10];       // grow-only device scratch: 0 comp, 1 tabs, 2 sym, 3 out, 4 cand, 5 ncand, 6 chain, 7 gmap, 8 gwin, 9 text
    char err[256];
};

static int ski_fail(sk_inflater *f, int code, const char *what, hipError_t e)
{
    snprintf(f->err, sizeof f->err, "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
    return code;
}
#define SKI_HIP(f, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return ski_fail((f), SK_E_HIP, #call, e_); } while (0)

static int ski_room(sk_inflater *f, int i, size_t need)
{
    if (need <= f->cap[i]) return SK_OK;
    if (f->buf[i]) { SKI_HIP(f, hipStreamSynchronize(f->stream)); (void)hipFree(f->buf[i]); f->buf[i] = NULL; f->cap[i] = 0; }
    const size_t want = need + need / 4 + 4096;
    SKI_HIP(f, hipMalloc(&f->buf[i], want));
    f->cap[i] = want;
    return SK_OK;
}

extern "C" int sk_inflater_create(int device, sk_inflater **out)
{
    if (!out) return SK_E_ARG;
    *out = NULL;
    if (hipSetDevice(device) != hipSuccess) return SK_E_NODEVICE;
    sk_inflater *f = new (std::nothrow) sk_inflater();
    if (!f) return SK_E_NOMEM;
    memset(f, 0, sizeof *f);
    f->device = device;
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) { delete f; return SK_E_HIP; }
    if (hipEventCreateWithFlags(&f->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(f->stream); delete f; return SK_E_HIP; }
    static bool attrs = false;
    if (!attrs) {
        (void)hipFuncSetAttribute((const void *)tails_compose, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WINDOW * 4));
        (void)hipFuncSetAttribute((const void *)tails_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WINDOW * 2));
        (void)hipFuncSetAttribute((const void *)resolve_translate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WINDOW * 2));
        attrs = true;
    }
    *out = f;
    return SK_OK;
}

extern "C" void sk_inflater_destroy(sk_inflater *f)
{
    if (!f) return;
    hipSetDevice(f->device);
    hipStreamSynchronize(f->stream);
    for (int i = 0; i < 10; i++) hipFree(f->buf[i]);
    hipEventDestroy(f->done);
    hipStreamDestroy(f->stream);
    delete f;
}

extern "C" const char *sk_inflater_error(const sk_inflater *f) { return f ? f->err : "no inflater"; }

// How many bytes the one member of this gzip file says it holds (ISIZE), or 0 if the file is nothing this path takes (then: the host).
extern "C" uint64_t sk_inflate_gz_size(const uint8_t *m, uint64_t n)
{
    if (!m || n < 32 || n > SKI_MAX_GZ || m[0] != 0x1f || m[1] != 0x8b || m[2] != 8 || (m[3] & 0xE0)) return 0;
    const uint32_t isize = (uint32_t)m[n - 4] | ((uint32_t)m[n - 3] << 8) | ((uint32_t)m[n - 2] << 16) | ((uint32_t)m[n - 1] << 24);
    if ((uint64_t)isize < n / 2 || (uint64_t)isize > n * 7ull) return 0;   // (a second member, or more than 4 GiB of text, or a ratio the buffers are not sized for)
    return isize;
}

// The whole member into host_text (page-locked, >= sk_inflate_gz_size() bytes).  SK_OK and *crc_ok_to_check = the trailer's CRC-32, or
// SK_E_UNSUPPORTED (nothing is promised about host_text), or a device error.
extern "C" int sk_inflate_gz(sk_inflater *f, const uint8_t *m, uint64_t ncomp, uint8_t *host_text, uint64_t host_cap, uint64_t *text_len, uint32_t *trailer_crc)
{
    if (!f || !m || !host_text || !text_len) return SK_E_ARG;
    const uint64_t want_len = sk_inflate_gz_size(m, ncomp);
    if (!want_len || want_len > host_cap) return SK_E_UNSUPPORTED;
    size_t hdr = 10;
    if (m[3] & 4) { if (hdr + 2 > ncomp) return SK_E_UNSUPPORTED; hdr += 2 + ((size_t)m[hdr] | ((size_t)m[hdr + 1] << 8)); }
    if (m[3] & 8) { while (hdr < ncomp && m[hdr]) hdr++; hdr++; }
    if (m[3] & 16) { while (hdr < ncomp && m[hdr]) hdr++; hdr++; }
    if (m[3] & 2) hdr += 2;
    if (hdr + 18 > ncomp) return SK_E_UNSUPPORTED;
    if (trailer_crc) *trailer_crc = (uint32_t)m[ncomp - 8] | ((uint32_t)m[ncomp - 7] << 8) | ((uint32_t)m[ncomp - 6] << 16) | ((uint32_t)m[ncomp - 5] << 24);
    const uint32_t seg_bytes = SKI_SEG_BYTES;
    const uint64_t first_bit = hdr * 8, total_bits = (ncomp - 8) * 8;
    const uint32_t nseg = (uint32_t)((ncomp - 8 - hdr + seg_bytes - 1) / seg_bytes);
    const uint32_t cap = seg_bytes * 8u + 66000u;
    const size_t nwords = (ncomp + 3) / 4;
    int rc;
    SKI_HIP(f, hipSetDevice(f->device));
    if ((rc = ski_room(f, 0, nwords * 4 + 16)) || (rc = ski_room(f, 1, (size_t)nseg * (NLIT + NDIST) * 4)) || (rc = ski_room(f, 2, (size_t)nseg * cap * 2)) ||
        (rc = ski_room(f, 3, (size_t)nseg * sizeof(seg_out))) || (rc = ski_room(f, 4, (size_t)nseg * CANDMAX * 4)) || (rc = ski_room(f, 5, (size_t)nseg * 4))) return rc;
    uint32_t *d_comp = (uint32_t *)f->buf[0], *d_tabs = (uint32_t *)f->buf[1], *d_cand = (uint32_t *)f->buf[4], *d_ncand = (uint32_t *)f->buf[5];
    uint16_t *d_sym = (uint16_t *)f->buf[2];
    seg_out *d_out = (seg_out *)f->buf[3];
    SKI_HIP(f, hipMemsetAsync((uint8_t *)d_comp + (nwords - 1) * 4, 0, 20, f->stream));
    SKI_HIP(f, hipMemcpyAsync(d_comp, m, ncomp, hipMemcpyHostToDevice, f->stream));
    SKI_HIP(f, hipMemsetAsync(d_ncand, 0, (size_t)nseg * 4, f->stream));
    hipLaunchKernelGGL(find_candidates, dim3((nseg + 3) / 4), dim3(256), 0, f->stream, (const uint32_t *)d_comp, (uint64_t)nwords, total_bits, seg_bytes, (uint64_t)hdr, nseg, d_cand, d_ncand);
    hipLaunchKernelGGL(validate, dim3((nseg + 3) / 4), dim3(256), 0, f->stream, (const uint32_t *)d_comp, (uint64_t)nwords, seg_bytes, (uint64_t)hdr, nseg, d_cand, (const uint32_t *)d_ncand);
    hipLaunchKernelGGL(spec_decode, dim3((nseg + 63) / 64), dim3(64), 0, f->stream, (const uint32_t *)d_comp, (uint64_t)nwords, first_bit, total_bits, seg_bytes, (uint64_t)hdr, nseg,
                       (const uint32_t *)d_cand, (const uint32_t *)d_ncand, d_tabs, d_sym, cap, d_out);
    hipLaunchKernelGGL(extend, dim3((nseg + 63) / 64), dim3(64), 0, f->stream, (const uint32_t *)d_comp, (uint64_t)nwords, nseg, d_tabs, d_sym, cap, d_out);
    std::vector<seg_out> out(nseg);
    SKI_HIP(f, hipMemcpyAsync(out.data(), d_out, nseg * sizeof(seg_out), hipMemcpyDeviceToHost, f->stream));
    SKI_HIP(f, hipEventRecord(f->done, f->stream));
    SKI_HIP(f, hipEventSynchronize(f->done));               // (a blocking wait: the feeding thread sleeps, the host's decode threads keep their CPUs)
    SKI_HIP(f, hipGetLastError());
    // the chain
    std::vector<chain_seg> ch;
    ch.reserve(nseg);
    uint64_t off = 0;
    int last = -1;
    for (uint32_t k = 0; k < nseg; k++) {
        if (!(out[k].flags & 1)) continue;
        if (last < 0 ? (k != 0 || out[k].start_bit != first_bit) : out[k].start_bit != out[last].end_bit) return SK_E_UNSUPPORTED;
        chain_seg c; c.seg = k; c.len = out[k].n; c.off = off;
        ch.push_back(c);
        off += out[k].n;
        last = (int)k;
    }
    if (last < 0 || !(out[last].flags & 2) || off != want_len) return SK_E_UNSUPPORTED;
    const uint32_t nch = (uint32_t)ch.size(), ngroups = (nch + GROUP - 1) / GROUP;
    if ((rc = ski_room(f, 6, nch * sizeof(chain_seg))) || (rc = ski_room(f, 7, (size_t)ngroups * WINDOW * 2)) || (rc = ski_room(f, 8, (size_t)ngroups * WINDOW)) || (rc = ski_room(f, 9, off + 16))) return rc;
    SKI_HIP(f, hipMemcpyAsync(f->buf[6], ch.data(), nch * sizeof(chain_seg), hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(tails_compose, dim3(ngroups), dim3(1024), WINDOW * 4, f->stream, (const chain_seg *)f->buf[6], nch, (const uint16_t *)d_sym, cap, (uint16_t *)f->buf[7]);
    hipLaunchKernelGGL(tails_chain, dim3(1), dim3(1024), WINDOW * 2, f->stream, (const uint16_t *)f->buf[7], ngroups, (uint8_t *)f->buf[8]);
    hipLaunchKernelGGL(resolve_translate, dim3(ngroups), dim3(1024), WINDOW * 2, f->stream, (const chain_seg *)f->buf[6], nch, (const uint16_t *)d_sym, cap, (const uint8_t *)f->buf[8], (uint8_t *)f->buf[9]);
    SKI_HIP(f, hipMemcpyAsync(host_text, f->buf[9], off, hipMemcpyDeviceToHost, f->stream));
    SKI_HIP(f, hipEventRecord(f->done, f->stream));
    SKI_HIP(f, hipEventSynchronize(f->done));               // (a blocking wait: the feeding thread sleeps, the host's decode threads keep their CPUs)
    SKI_HIP(f, hipGetLastError());
    *text_len = off;
    return SK_OK;
}
