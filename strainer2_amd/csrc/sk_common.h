/* sk_common.h -- definitions shared by the host (C) and device (HIP) halves of libstrainer_kmer.
 *
 * Packed key convention (include/strainer_kmer.h): 31 bases, 2 bits each, A=0 C=1 G=2 T=3,
 * first base in bits 61..60.  With that code the integer order of packed keys equals the
 * signed-char ASCII order the reference compares in (A<C<G<T), so the reference's
 * "lexicographically larger of window and reverse complement, forward on ties"
 * (src/genome_compare.c:1100-1141) is max(fwd, rc) on integers.
 */
#ifndef SK_COMMON_H
#define SK_COMMON_H
#include <stdint.h>

#if defined(__HIPCC__)
#define SK_HD __host__ __device__ __forceinline__
#else
#define SK_HD static inline
#endif

#define SK_KMASK62   0x3FFFFFFFFFFFFFFFull
#define SK_EMPTY64   0xFFFFFFFFFFFFFFFFull
#define SK_OVERLAP   30u                 /* k-1 bytes shared by consecutive pieces of a cut record */

/* 2-bit code of an (any-case) base byte; garbage for other bytes (callers gate on sk_is_acgt) */
SK_HD uint32_t sk_code(uint32_t b)
{
    uint32_t x = (b >> 1) & 3u;          /* A0 C1 T2 G3 */
    return x ^ (x >> 1);                 /* A0 C1 G2 T3 */
}

SK_HD int sk_is_acgt(uint32_t b)
{
    uint32_t u = b & 0xDFu;
    return (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
}

/* bytes that can never be inside a counted window: N (src/genome_compare.c:210,219,1007),
 * the record separator, and NUL (C-string end in the reference) */
SK_HD int sk_is_hard_break(uint32_t b)
{
    return ((b & 0xDFu) == 'N') | (b == '\n') | (b == 0u);
}

/* hash of a packed canonical key for the HOST builder's de-duplication set (the reference's djb2
 * only matters for row ORDER and is replayed separately; the device table uses sk_khash) */
SK_HD uint32_t sk_hash62(uint64_t key)
{
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = lo ^ (hi * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

/* ---- minimizers (prefilter) ----------------------------------------------------------------
 * Every 31-mer window contains w = 16 overlapping 16-mers (m = 16, 32 bits packed).  The scan
 * computes the FORWARD-strand minimizer hash of each window, mz = min over those of
 * sk_mhash(16-mer); no reverse complement is involved there.  Strand symmetry comes from the
 * build side instead: for every strain key K both mz(K) and mz(revcomp K) go into the filter, so
 * a read window that equals K in either orientation finds its own forward minimizer present.
 * Consecutive windows mostly share their minimizer: one filter lookup per run of ~7 windows. */
SK_HD uint32_t sk_mhash(uint32_t f16)
{
    /* one multiply; the multiplier is EVEN so bit 0 of every hash is 0 and 0xFFFFFFFF (the scan's
     * "dead stretch" marker) is never a hash */
    return f16 * 0x9E3779B2u;
}

SK_HD uint32_t sk_khash(uint64_t key)
{
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t x = lo ^ (hi * 0x85EBCA77u);
    x ^= x >> 15;
    return x * 0x9E3779B1u;                 /* use the HIGH bits */
}

/* minimizer filter: block index and the four test bits of a minimizer hash */
SK_HD uint32_t sk_filter_block(uint32_t mz, uint32_t shift) { return (mz * 0x9E3779B1u) >> shift; }
SK_HD uint32_t sk_filter_bits(uint32_t mz) { return mz * 0x85EBCA77u; }

/* ---- grid 16-mers (prefilter of the grid kernel) ----------------------------------------------
 * Cut the stream into 16-base chunks at multiples of 16.  Every 31-base window contains exactly one
 * whole chunk (the one starting in its first 16 bases), so the windows are partitioned by "their"
 * chunk, 16 windows each.  If a window is a strain k-mer, its chunk is a 16-mer of the strain (in one
 * orientation or the other).  The filter therefore holds the canonical form (min of the packed 16-mer
 * and its reverse complement) of EVERY 16-mer of every strain key; a chunk that is not in it rules out
 * its 16 windows with one lookup per 16 bases and no per-base work at all. */
SK_HD uint32_t sk_revcomp16(uint32_t x)                /* 16 packed bases */
{
    uint32_t y = 0;
    for (int i = 0; i < 16; i++) { y = (y << 2) | (3u - (x & 3u)); x >>= 2; }
    return y;
}
SK_HD uint32_t sk_gmix(uint32_t canon16) { return canon16 ^ (canon16 >> 15); }
/* level 1 (L2-resident) and level 2 (large) filters: 64-bit blocks, two bits in each 32-bit half */
/* level 1 may have any number of blocks (it is sized to what the L2 keeps, not to a power of two):
 * block = high word of hash x nblocks */
SK_HD uint32_t sk_grid1_block(uint32_t g, uint32_t nblocks) { return (uint32_t)(((uint64_t)(g * 0x9E3779B1u) * nblocks) >> 32); }
SK_HD uint32_t sk_grid1_bits(uint32_t g) { return g * 0x85EBCA77u; }
SK_HD uint32_t sk_grid2_block(uint32_t g, uint32_t shift) { return (g * 0xC2B2AE3Du) >> shift; }
SK_HD uint32_t sk_grid2_bits(uint32_t g) { return g * 0x27D4EB2Fu; }

/* reverse complement of a packed 31-mer */
SK_HD uint64_t sk_revcomp62(uint64_t key)
{
    uint64_t r = 0;
    for (int i = 0; i < 31; i++) { r = (r << 2) | (3u - (key & 3u)); key >>= 2; }
    return r;
}

/* forward minimizer hash of a packed 31-mer (build side; the scan kernel slides it) */
SK_HD uint32_t sk_minimizer62(uint64_t key)
{
    uint32_t mz = 0xFFFFFFFFu;
    for (int i = 0; i < 16; i++) {
        const uint32_t h = sk_mhash((uint32_t)(key >> (2 * (15 - i))));
        mz = h < mz ? h : mz;
    }
    return mz;
}

/* first table slot of a key: uniform over the table (bits 8..31 of the k-mer hash); linear
 * probing from there.  (Placing table lines by minimizer was tried and rejected: keys arrive in
 * clumps of ~8 per minimizer and linear-probe runs get long: 3.7x slower end to end.) */
SK_HD uint32_t sk_slot0(uint32_t kh, uint32_t mask)
{
    /* bits 8..31 of the hash, and -- for tables of more than 2^24 slots (strains beyond ~8 M keys) -- its low byte on
     * top of them: with 24 bits alone every key of a bigger table starts in the first 2^24 slots, and a 20 M-key strain
     * (more keys than home slots) turned linear probing into a walk of millions of slots per key */
    return ((kh >> 8) | (kh << 24)) & mask;
}

/* FNV-1a over the 31 bytes of a wide key */
SK_HD uint32_t sk_hash_wide(const char *k31)
{
    uint32_t h = 2166136261u;
    for (int i = 0; i < 31; i++) { h ^= (uint8_t)k31[i]; h *= 16777619u; }
    return h;
}

/* C-locale toupper, the only case mapping the reference applies (src/BIO_sequence.c:228-234) */
SK_HD uint32_t sk_upper(uint32_t b) { return (b >= 'a' && b <= 'z') ? b - 32u : b; }

/* The reference's nucleotide complement map (src/BIO_sequence.c:203-213), as data: every byte
 * maps to (char)-1 except the IUPAC letters and a few symbols.  Reproduced quirks: upper-case
 * K maps to '.', U maps to A.  Bytes >= 0x80 (negative subscript in the reference: undefined)
 * map to -1 here. */
static inline void sk_fill_complement(signed char t[256])
{
    static const char from[] = "-.^ATCGBVDHKMNRYSUWXatcgbvdhkmnrysuwx";
    static const char to[]   = "-.^TAGCVBHD.KNYRSAWXtagcvbhdmknyrsawx";
    int i;
    for (i = 0; i < 256; i++) t[i] = -1;
    for (i = 0; from[i]; i++) t[(unsigned char)from[i]] = (signed char)to[i];
}

#endif
