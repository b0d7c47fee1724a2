/* sk_pack.h -- the host-side 2-bit pre-pack of a record stream (internal; SURVEY 8(f1): "optional on-host 2-bit pre-pack to
 * quarter PCIe bytes").
 *
 * Both programs' passes over plain text were bound by the PCIe link at the end of round 4 (112 GB of record bytes per configs[2]
 * job at 51 GB/s; waits that sleep instead of spinning free a quarter of the CPU time and change nothing).  What the scan
 * kernel makes of a 16-byte chunk of the stream in its phase 1 (sk_decode16, sk_dev_scan.hip.h) can be made on the host as well:
 *   code word  32 bits, the chunk's sixteen 2-bit codes, first byte highest (A 0, C 1, G 2, T 3 in either case; 0 for any other byte)
 *   mask       16 bits, bit i <=> byte i is no A/C/G/T
 * -- 6 bytes a chunk instead of 16.  A batch that holds a byte which is neither A/C/G/T, N/n nor '\n' (an IUPAC letter, 'U', a
 * '\r' ...: bytes only the exact byte-string kernel can judge, src/BIO_sequence.c:203-213) is NOT packed: *odd comes back set and
 * the caller sends the bytes as they are.  Bytes of the last chunk beyond the stream's end are "no A/C/G/T".
 * Layout of `packed`: (nbytes + 15) / 16 code words, then as many masks.  AVX2 + BMI2 where the CPU has them (32 bytes a step, the codes
 * gathered by pext), a table otherwise; chosen once, at run time.  (A 64-byte AVX-512 form -- mask registers and pdep -- was written and
 * measured on the EPYC 9575F: 8.9 s against 7.5 s of packing per configs[2] job; the pass over a 32 MiB chunk is bound by memory, not by
 * instructions.  Taken out.) */
#ifndef SK_PACK_H
#define SK_PACK_H
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

static inline void skp_pack_plain(const uint8_t *b, uint64_t n, uint32_t *codes, uint16_t *inv, int *odd)
{
    static const int8_t lut[256] = {
#define X4 -2, -2, -2, -2
#define X16 X4, X4, X4, X4
        -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -1, -2, -2, -2, -2, -2,  X16,  X16,  X16,
        /* 0x40 */ -2, 0, -2, 1, -2, -2, -2, 2, -2, -2, -2, -2, -2, -2, -1, -2,   /* @ A B C D E F G H I J K L M N O */
        /* 0x50 */ -2, -2, -2, -2, 3, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, /* P Q R S T ... */
        /* 0x60 */ -2, 0, -2, 1, -2, -2, -2, 2, -2, -2, -2, -2, -2, -2, -1, -2,
        /* 0x70 */ -2, -2, -2, -2, 3, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2,
        X16, X16, X16, X16, X16, X16, X16, X16
#undef X16
#undef X4
    };                                                   /* 0..3: the base's code; -1: N, n or '\n'; -2: a byte for the byte-string kernel */
    const uint64_t nch = (n + 15u) >> 4;
    uint64_t g;
    int any_odd = 0;
    for (g = 0; g < nch; g++) {
        uint32_t c = 0, m = 0;
        unsigned i;
        for (i = 0; i < 16; i++) {
            const uint64_t at = g * 16u + i;
            const int v = at < n ? lut[b[at]] : -1;
            c <<= 2;
            if (v >= 0) c |= (uint32_t)v; else { m |= 1u << i; any_odd |= v == -2; }
        }
        codes[g] = c;
        inv[g] = (uint16_t)m;
    }
    if (any_odd) *odd = 1;
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2")))
static inline void skp_pack_avx2(const uint8_t *b, uint64_t n, uint32_t *codes, uint16_t *inv, int *odd)
{
    const uint64_t full = n >> 5;                        /* steps of 32 bytes = two chunks */
    const __m256i up = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'),
                  cT = _mm256_set1_epi8('T'), cN = _mm256_set1_epi8('N'), nl = _mm256_set1_epi8('\n'), three = _mm256_set1_epi8(3);
    __m256i oddacc = _mm256_setzero_si256();
    uint64_t s;
    for (s = 0; s < full; s++) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(const void *)(b + 32 * s));
        const __m256i u = _mm256_and_si256(x, up);
        const __m256i isT = _mm256_cmpeq_epi8(u, cT), isG = _mm256_cmpeq_epi8(u, cG);
        const __m256i valid = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)), _mm256_or_si256(isG, isT));
        const __m256i fine = _mm256_or_si256(valid, _mm256_or_si256(_mm256_cmpeq_epi8(u, cN), _mm256_cmpeq_epi8(x, nl)));
        /* (u >> 1) & 3: A 0, C 1, G 3, T 2 -> G and T trade places: x ^ (x >> 1) on two bits */
        __m256i c = _mm256_and_si256(_mm256_srli_epi16(u, 1), three);
        c = _mm256_xor_si256(c, _mm256_and_si256(_mm256_srli_epi16(c, 1), _mm256_set1_epi8(1)));
        c = _mm256_and_si256(c, valid);
        oddacc = _mm256_or_si256(oddacc, _mm256_andnot_si256(fine, _mm256_set1_epi8(1)));
        {
            const uint32_t vm = (uint32_t)_mm256_movemask_epi8(valid);
            uint64_t q[4];
            _mm256_storeu_si256((__m256i *)(void *)q, c);
            codes[2 * s]     = ((uint32_t)_pext_u64(__builtin_bswap64(q[0]), 0x0303030303030303ull) << 16) | (uint32_t)_pext_u64(__builtin_bswap64(q[1]), 0x0303030303030303ull);
            codes[2 * s + 1] = ((uint32_t)_pext_u64(__builtin_bswap64(q[2]), 0x0303030303030303ull) << 16) | (uint32_t)_pext_u64(__builtin_bswap64(q[3]), 0x0303030303030303ull);
            inv[2 * s]     = (uint16_t)~vm;
            inv[2 * s + 1] = (uint16_t)(~vm >> 16);
        }
    }
    if (!_mm256_testz_si256(oddacc, oddacc)) *odd = 1;
    if (n & 31u) skp_pack_plain(b + 32 * full, n & 31u, codes + 2 * full, inv + 2 * full, odd);
}
#endif

typedef void (*skp_pack_fn)(const uint8_t *, uint64_t, uint32_t *, uint16_t *, int *);
static inline skp_pack_fn skp_pack_pick(void)
{
#if defined(__x86_64__)
    const char *e = getenv("SK_PACK_SIMD");              /* 0: the table (tests run both) */
    __builtin_cpu_init();
    if (!(e && e[0] == '0') && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) return skp_pack_avx2;
#endif
    return skp_pack_plain;
}
#endif
