/* sk_gzfast.h -- a from-scratch gzip/DEFLATE (RFC 1951/1952) decoder for the ingest threads.
 *
 * Inflate is what the end-to-end rate of this whole program is made of once the scan runs on the GPU
 * (zlib's gzread: ~0.27 Gbase/s per thread on FASTQ).  This decoder does the same job about twice as
 * fast by the usual means: the compressed file is mapped into memory whole (no input refill logic in the
 * inner loop), a 64-bit bit buffer is refilled with one unaligned load, literal/length and distance codes
 * are decoded through one table lookup (11- and 8-bit primary tables, subtables for longer codes), matches
 * are copied 8 bytes at a time, and output goes to a large buffer that is handed to the consumer in
 * pieces of a few MB (the last 32 KiB stay in front of it as the match history).
 *
 * Contract (what the callers rely on, checked against zlib in tests/test_gzfast.py):
 *   skz_decode_file(path, sink, user)
 *     returns SKZ_NOT_GZIP before any output when the file does not start with a gzip member (the caller
 *     then takes its zlib route, which also passes plain files through);
 *     otherwise feeds the decompressed bytes of all concatenated members, in order, to sink(); data after
 *     the last well-formed member is ignored, as gzread does; on a corrupt or truncated stream the bytes
 *     decoded up to the fault have been delivered and SKZ_CORRUPT is returned (gzread delivers them too
 *     before reporting its error); the CRC-32 and length of every member are verified.
 *     A non-zero return of sink() stops the decoding (SKZ_STOPPED).
 * Host only, C99, no dependencies beyond libc.
 */
#ifndef SK_GZFAST_H
#define SK_GZFAST_H
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

enum { SKZ_OK = 0, SKZ_OPEN = -1, SKZ_NOT_GZIP = 1, SKZ_CORRUPT = 2, SKZ_STOPPED = 3 };
typedef int (*skz_sink)(void *user, const unsigned char *data, size_t n);

#ifndef SKZ_LITLEN_BITS
#define SKZ_LITLEN_BITS 11
#endif
#ifndef SKZ_DIST_BITS
#define SKZ_DIST_BITS   8
#endif
#define SKZ_WINDOW      32768u
#define SKZ_OUT_CHUNK   (4u << 20)

/* table entry: bits 0..3 code length still to consume at this level, bits 4..7 kind, bits 8..15 number of
 * extra bits (length/distance) or of subtable index bits, bits 16..31 literal / base value / subtable start */
enum { SKZ_K_LIT = 0, SKZ_K_LIT2 = 1, SKZ_K_LEN = 2, SKZ_K_EOB = 3, SKZ_K_SUB = 4, SKZ_K_BAD = 5, SKZ_K_DIST = 6 };
#define SKZ_ENTRY(val, extra, kind, nbits) (((uint32_t)(val) << 16) | ((uint32_t)(extra) << 8) | ((uint32_t)(kind) << 4) | (uint32_t)(nbits))

typedef struct {
    uint32_t litlen[(1u << SKZ_LITLEN_BITS) + 286 * 16];   /* primary + subtables (one of 2^(15-11) per long code at most) */
    uint32_t dist[(1u << SKZ_DIST_BITS) + 30 * 128];
    uint8_t  lens[320]; unsigned nlit, ndist;              /* the code lengths the tables were built from (for skz_tail) */
} skz_tables;

static const uint16_t skz_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t  skz_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t skz_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t  skz_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t skz_rev(uint32_t code, int len)
{
    uint32_t r = 0;
    int i;
    for (i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

/* what symbol `sym` of the literal/length (is_dist = 0) or distance alphabet decodes to */
static inline uint32_t skz_symbol_entry(int is_dist, unsigned sym, int nbits)
{
    if (is_dist) return sym < 30 ? SKZ_ENTRY(skz_dist_base[sym], skz_dist_extra[sym], SKZ_K_DIST, nbits) : SKZ_ENTRY(0, 0, SKZ_K_BAD, nbits);
    if (sym < 256) return SKZ_ENTRY(sym, 0, SKZ_K_LIT, nbits);
    if (sym == 256) return SKZ_ENTRY(0, 0, SKZ_K_EOB, nbits);
    if (sym < 286) return SKZ_ENTRY(skz_len_base[sym - 257], skz_len_extra[sym - 257], SKZ_K_LEN, nbits);
    return SKZ_ENTRY(0, 0, SKZ_K_BAD, nbits);
}

/* Canonical Huffman decode table from code lengths (0 = unused).  Codes up to `tbits` long fill the primary
 * table directly (replicated); longer ones go through a subtable per distinct low-`tbits` prefix.  Returns 0,
 * or -1 for an over-subscribed code.  Incomplete codes are accepted (unused entries decode as SKZ_K_BAD), as
 * zlib accepts the one-distance-code case; an all-zero distance alphabet is accepted too. */
static int skz_build(uint32_t *table, int tbits, size_t table_cap, const uint8_t *lens, unsigned nsym, int is_dist)
{
    unsigned count[16] = {0}, next[16], next2[16], sym, len;
    uint32_t code = 0, left = 1;
    size_t used = (size_t)1 << tbits, i;
    uint8_t sub_max[1u << SKZ_LITLEN_BITS];           /* per primary slot: longest code through it */
    for (sym = 0; sym < nsym; sym++) count[lens[sym]]++;
    count[0] = 0;
    for (len = 1; len <= 15; len++) {
        left <<= 1;
        if (count[len] > left) return -1;
        left -= count[len];
    }
    for (len = 1; len <= 15; len++) { code = (code + count[len - 1]) << 1; next[len] = next2[len] = code; }
    for (i = 0; i < ((size_t)1 << tbits); i++) table[i] = SKZ_ENTRY(0, 0, SKZ_K_BAD, 1);
    memset(sub_max, 0, (size_t)1 << tbits);
    /* pass 1: short codes, and the depth of every subtable */
    for (sym = 0; sym < nsym; sym++) {
        uint32_t r;
        len = lens[sym];
        if (!len) continue;
        r = skz_rev(next[len]++, (int)len);              /* codes of one length are handed out in symbol order */
        if ((int)len <= tbits) {
            const uint32_t e = skz_symbol_entry(is_dist, sym, (int)len);
            for (i = r; i < ((size_t)1 << tbits); i += (size_t)1 << len) table[i] = e;
        } else {
            const uint32_t p = r & ((1u << tbits) - 1u);
            if (len > sub_max[p]) sub_max[p] = (uint8_t)len;
        }
    }
    /* allocate the subtables */
    for (i = 0; i < ((size_t)1 << tbits); i++)
        if (sub_max[i]) {
            const unsigned sbits = (unsigned)sub_max[i] - (unsigned)tbits;
            size_t j;
            if (used + ((size_t)1 << sbits) > table_cap || used > 0xFFFFu) return -1;
            table[i] = SKZ_ENTRY(used, sbits, SKZ_K_SUB, tbits);
            for (j = 0; j < ((size_t)1 << sbits); j++) table[used + j] = SKZ_ENTRY(0, 0, SKZ_K_BAD, 1);
            used += (size_t)1 << sbits;
        }
    /* pass 2: long codes into their subtables */
    for (sym = 0; sym < nsym; sym++) {
        uint32_t r, p, start, sbits, e;
        len = lens[sym];
        if ((int)len <= tbits) continue;
        r = skz_rev(next2[len]++, (int)len);
        p = r & ((1u << tbits) - 1u);
        start = table[p] >> 16;
        sbits = (table[p] >> 8) & 0xFFu;
        e = skz_symbol_entry(is_dist, sym, (int)len - tbits);
        for (i = r >> tbits; i < ((size_t)1 << sbits); i += (size_t)1 << (len - (unsigned)tbits)) table[start + i] = e;
    }
    /* literal pairs: where the bits of a primary index spell one literal and then, completely, a second one,
     * the entry yields both (sequence data is mostly literals with codes of 2-6 bits) */
    if (!is_dist) {
        uint32_t prim[1u << SKZ_LITLEN_BITS];
        memcpy(prim, table, sizeof prim);
        for (i = 0; i < ((size_t)1 << tbits); i++) {
            const uint32_t e1 = prim[i], l1 = e1 & 15u;
            uint32_t e2;
            if (((e1 >> 4) & 15u) != SKZ_K_LIT || l1 >= (uint32_t)tbits) continue;
            e2 = prim[i >> l1];                         /* (the unknown high bits read as zeros) */
            if (((e2 >> 4) & 15u) != SKZ_K_LIT || (e2 & 15u) > (uint32_t)tbits - l1) continue;
            table[i] = SKZ_ENTRY((e1 >> 16) | ((e2 >> 16) << 8), 0, SKZ_K_LIT2, l1 + (e2 & 15u));
        }
    }
    return 0;
}

typedef struct {
    const unsigned char *in, *in_end;
    uint64_t bitbuf; unsigned bitcnt;
    unsigned char *out_base, *out, *out_flushed, *out_end;     /* out_base..: [32 KiB history][chunk][slack] */
    skz_sink sink; void *user;
    uint32_t crc; uint64_t total;
    int stopped;
    const unsigned char *data;                                 /* start of the mapped file (bit positions count from here) */
    uint64_t stop_bit;                                         /* skz_inflate returns 2 at the first block boundary at or past this bit (0: none) */
    int skip_crc;
} skz_stream;

static uint32_t skz_crc_tab[8][256];
static pthread_once_t skz_crc_once = PTHREAD_ONCE_INIT;
static uint32_t skz_crc32_table(uint32_t crc, const unsigned char *p, size_t n);
#if defined(__x86_64__)
static uint32_t skz_crc32_clmul(uint32_t crc, const unsigned char *buf, size_t n);
static int skz_crc_clmul_ok;
#endif
static void skz_crc_init(void)
{
    uint32_t i, j;
    for (i = 0; i < 256; i++) {
        uint32_t c = i;
        for (j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        skz_crc_tab[0][i] = c;
    }
    for (i = 0; i < 256; i++)
        for (j = 1; j < 8; j++) skz_crc_tab[j][i] = (skz_crc_tab[j - 1][i] >> 8) ^ skz_crc_tab[0][skz_crc_tab[j - 1][i] & 255u];
#if defined(__x86_64__)
    if (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) {
        unsigned char t[64 * 5 + 16];
        uint32_t seed = 0x12345678u, ok = 1, n;
        for (i = 0; i < sizeof t; i++) { seed = seed * 1664525u + 1013904223u; t[i] = (unsigned char)(seed >> 24); }
        for (n = 64; n <= sizeof t; n += 16)
            ok &= (uint32_t)(~skz_crc32_clmul(~0x9ABCDEF0u, t, n) == skz_crc32_table(0x9ABCDEF0u, t, n));
        skz_crc_clmul_ok = (int)ok;
    }
#endif
}
static uint32_t skz_crc32_table(uint32_t crc, const unsigned char *p, size_t n)
{
    crc = ~crc;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4); memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = skz_crc_tab[7][a & 255u] ^ skz_crc_tab[6][(a >> 8) & 255u] ^ skz_crc_tab[5][(a >> 16) & 255u] ^ skz_crc_tab[4][a >> 24] ^
              skz_crc_tab[3][b & 255u] ^ skz_crc_tab[2][(b >> 8) & 255u] ^ skz_crc_tab[1][(b >> 16) & 255u] ^ skz_crc_tab[0][b >> 24];
        p += 8; n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ skz_crc_tab[0][(crc ^ *p++) & 255u];
    return ~crc;
}

#if defined(__x86_64__)
#include <immintrin.h>
/* CRC-32 by carry-less multiplication (folding 64 bytes per step, then Barrett reduction; constants of the
 * reflected polynomial 0xEDB88320 as in the Intel white paper).  n >= 64 and a multiple of 16; `crc` and the
 * result are the raw (inverted) register.  Used only after skz_crc_init has checked it against the table
 * version on this machine. */
__attribute__((target("pclmul,sse4.1")))
static uint32_t skz_crc32_clmul(uint32_t crc, const unsigned char *buf, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64; n -= 64;
    while (n >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i *)(buf + 0x00)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i *)(buf + 0x10)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i *)(buf + 0x20)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i *)(buf + 0x30)));
        buf += 64; n -= 64;
    }
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (n >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; n -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

static uint32_t skz_crc32(uint32_t crc, const unsigned char *p, size_t n)
{
#if defined(__x86_64__)
    if (skz_crc_clmul_ok && n >= 64) {
        const size_t body = n & ~(size_t)15;
        crc = ~skz_crc32_clmul(~crc, p, body);
        p += body; n -= body;
    }
#endif
    return skz_crc32_table(crc, p, n);
}

/* hand everything decoded so far to the consumer; keep the last 32 KiB as history when the buffer is full */
static int skz_flush(skz_stream *s, int make_room)
{
    const size_t n = (size_t)(s->out - s->out_flushed);
    if (n) {
        if (!s->skip_crc) s->crc = skz_crc32(s->crc, s->out_flushed, n);
        s->total += n;
        if (!s->stopped && s->sink(s->user, s->out_flushed, n)) s->stopped = 1;
        s->out_flushed = s->out;
    }
    if (make_room && (size_t)(s->out - s->out_base) > SKZ_WINDOW) {
        memmove(s->out_base, s->out - SKZ_WINDOW, SKZ_WINDOW);
        s->out = s->out_flushed = s->out_base + SKZ_WINDOW;
    }
    return s->stopped;
}

/* bits: refill to at least 56 valid bits when the input allows; past the end zeros are supplied and the
 * overrun is detected by the callers through in > in_end */
#define SKZ_REFILL(s) do {                                                                     \
        if ((s)->in + 8 <= (s)->in_end) {                                                      \
            uint64_t w_; memcpy(&w_, (s)->in, 8);                                              \
            (s)->bitbuf |= w_ << (s)->bitcnt;                                                  \
            (s)->in += (63u - (s)->bitcnt) >> 3;                                               \
            (s)->bitcnt |= 56u;                                                                \
        } else {                                                                               \
            while ((s)->bitcnt <= 56u) {                                                       \
                const uint64_t b_ = (s)->in < (s)->in_end ? *(s)->in : 0u;                     \
                (s)->in++;                                                                     \
                (s)->bitbuf |= b_ << (s)->bitcnt; (s)->bitcnt += 8u;                           \
            }                                                                                  \
        }                                                                                      \
    } while (0)
#define SKZ_BITS(s, n)  ((uint32_t)((s)->bitbuf & (((uint64_t)1 << (n)) - 1u)))
#define SKZ_DROP(s, n)  do { (s)->bitbuf >>= (n); (s)->bitcnt -= (unsigned)(n); } while (0)

/* more input consumed than there was (bits still in the buffer are counted as not consumed) */
static inline int skz_overrun(const skz_stream *s) { return s->in > s->in_end + (s->bitcnt >> 3); }

/* 1 when the code lengths describe a complete prefix code (Kraft sum exactly 1) */
static int skz_complete(const uint8_t *lens, unsigned nsym)
{
    uint32_t sum = 0;
    unsigned i;
    for (i = 0; i < nsym; i++) if (lens[i]) sum += 1u << (15 - lens[i]);
    return sum == (1u << 15);
}

/* strict: also insist on what zlib insists on for a block it would accept -- a complete literal/length code and a
 * distance code that is complete or a single code (used by the block finder of sk_gzpar.h, where a header is a guess) */
static int skz_read_dynamic(skz_stream *s, skz_tables *t, int strict)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t lens[320], cl[19];
    uint32_t cltab[128 + 64];
    unsigned hlit, hdist, hclen, i, n;
    SKZ_REFILL(s);
    hlit = SKZ_BITS(s, 5) + 257; SKZ_DROP(s, 5);
    hdist = SKZ_BITS(s, 5) + 1; SKZ_DROP(s, 5);
    hclen = SKZ_BITS(s, 4) + 4; SKZ_DROP(s, 4);
    if (hlit > 286 || hdist > 30) return -1;
    memset(cl, 0, sizeof cl);
    for (i = 0; i < hclen; i++) {
        if ((i & 7u) == 0) SKZ_REFILL(s);
        cl[order[i]] = (uint8_t)SKZ_BITS(s, 3); SKZ_DROP(s, 3);
    }
    /* code-length alphabet: a small table, 7 bits cover every code (max length 7) */
    {
        unsigned count[8] = {0}, next[8], code = 0, len, sym;
        uint32_t left = 1;
        for (sym = 0; sym < 19; sym++) count[cl[sym]]++;
        count[0] = 0;
        for (len = 1; len <= 7; len++) { left <<= 1; if (count[len] > left) return -1; left -= count[len]; }
        for (len = 1; len <= 7; len++) { code = (code + count[len - 1]) << 1; next[len] = code; }
        for (i = 0; i < 128; i++) cltab[i] = 0xFFFFFFFFu;
        for (sym = 0; sym < 19; sym++) {
            uint32_t r, j;
            len = cl[sym];
            if (!len) continue;
            r = skz_rev(next[len]++, (int)len);
            for (j = r; j < 128; j += 1u << len) cltab[j] = (sym << 8) | len;
        }
    }
    n = 0;
    while (n < hlit + hdist) {
        uint32_t e, sym, rep, val;
        SKZ_REFILL(s);
        e = cltab[SKZ_BITS(s, 7)];
        if (e == 0xFFFFFFFFu) return -1;
        SKZ_DROP(s, e & 255u);
        sym = e >> 8;
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        if (sym == 16) { if (n == 0) return -1; val = lens[n - 1]; rep = 3 + SKZ_BITS(s, 2); SKZ_DROP(s, 2); }
        else if (sym == 17) { val = 0; rep = 3 + SKZ_BITS(s, 3); SKZ_DROP(s, 3); }
        else { val = 0; rep = 11 + SKZ_BITS(s, 7); SKZ_DROP(s, 7); }
        if (n + rep > hlit + hdist) return -1;
        while (rep--) lens[n++] = (uint8_t)val;
    }
    if (skz_overrun(s) || lens[256] == 0) return -1;
    if (strict) {
        unsigned nd = 0;
        for (i = 0; i < hdist; i++) nd += lens[hlit + i] != 0;
        if (!skz_complete(lens, hlit)) return -1;
        if (nd > 1 && !skz_complete(lens + hlit, hdist)) return -1;
    }
    if (skz_build(t->litlen, SKZ_LITLEN_BITS, sizeof t->litlen / sizeof t->litlen[0], lens, hlit, 0)) return -1;
    if (skz_build(t->dist, SKZ_DIST_BITS, sizeof t->dist / sizeof t->dist[0], lens + hlit, hdist, 1)) return -1;
    memcpy(t->lens, lens, hlit + hdist); t->nlit = hlit; t->ndist = hdist;
    return 0;
}

static void skz_fixed_tables(skz_tables *t)
{
    uint8_t lens[288 + 32];
    unsigned i;
    for (i = 0; i < 144; i++) lens[i] = 8;
    for (; i < 256; i++) lens[i] = 9;
    for (; i < 280; i++) lens[i] = 7;
    for (; i < 288; i++) lens[i] = 8;
    skz_build(t->litlen, SKZ_LITLEN_BITS, sizeof t->litlen / sizeof t->litlen[0], lens, 288, 0);
    memcpy(t->lens, lens, 288); t->nlit = 288;
    for (i = 0; i < 32; i++) lens[i] = 5;
    skz_build(t->dist, SKZ_DIST_BITS, sizeof t->dist / sizeof t->dist[0], lens, 32, 1);
    memcpy(t->lens + 288, lens, 32); t->ndist = 32;
}

/* ---- the last bytes of the input, bit by bit --------------------------------------------------------------
 * The table decoder above looks codes up in bits that may lie past the end of the input (zeros are supplied).
 * In a complete file that never matters; in a file that was cut short it decides how much text comes out, and
 * the reference (zlib's gzread under kseq) delivers exactly the symbols all of whose bits are there: a literal
 * when its code is complete, a match when its length code, length bits, distance code and distance bits all
 * are.  Within 16 bytes of the end skz_block therefore hands over to this decoder, which reads real bits only
 * (canonical decoding from the code lengths, one bit at a time -- a few hundred symbols per file at most). */
typedef struct { unsigned short count[16], symbol[288]; } skz_canon;

static void skz_canon_build(skz_canon *c, const uint8_t *lens, unsigned n)
{
    unsigned offs[16], i;
    memset(c->count, 0, sizeof c->count);
    for (i = 0; i < n; i++) c->count[lens[i]]++;
    c->count[0] = 0;
    offs[1] = 0;
    for (i = 1; i < 15; i++) offs[i + 1] = offs[i] + c->count[i];
    for (i = 0; i < n; i++) if (lens[i]) c->symbol[offs[lens[i]]++] = (unsigned short)i;
}

/* one real bit, -1 when the input is used up */
static inline int skz_real_bit(skz_stream *s)
{
    int b;
    if (!s->bitcnt) {
        if (s->in >= s->in_end) return -1;
        s->bitbuf = *s->in++; s->bitcnt = 8;
    }
    b = (int)(s->bitbuf & 1u);
    s->bitbuf >>= 1; s->bitcnt--;
    return b;
}
static inline int skz_real_bits(skz_stream *s, unsigned n, uint32_t *v)
{
    unsigned i;
    *v = 0;
    for (i = 0; i < n; i++) { const int b = skz_real_bit(s); if (b < 0) return -1; *v |= (uint32_t)b << i; }
    return 0;
}
/* the next symbol of a canonical code: >= 0, -1 input used up, -2 no such code */
static int skz_real_symbol(skz_stream *s, const skz_canon *c)
{
    int code = 0, first = 0, index = 0, len;
    for (len = 1; len <= 15; len++) {
        const int b = skz_real_bit(s), count = c->count[len];
        if (b < 0) return -1;
        code |= b;
        if (code - count < first) return c->symbol[index + (code - first)];
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return -2;
}

/* the rest of the current block from the state in *s (every bit in the bit buffer is a real one): 0 at the
 * end-of-block code, 1 stopped by the consumer, -1 when the input ends first or is damaged */
static int skz_tail(skz_stream *s, const skz_tables *t)
{
    skz_canon lit, dst;
    skz_canon_build(&lit, t->lens, t->nlit);
    skz_canon_build(&dst, t->lens + t->nlit, t->ndist);
    for (;;) {
        int sym;
        if ((size_t)(s->out_end - s->out) < 258 + 16 && skz_flush(s, 1)) return 1;
        sym = skz_real_symbol(s, &lit);
        if (sym < 0) return -1;
        if (sym < 256) { *s->out++ = (unsigned char)sym; continue; }
        if (sym == 256) return 0;
        if (sym >= 286) return -1;
        {
            uint32_t x, len, dist;
            int ds;
            unsigned char *d;
            if (skz_real_bits(s, skz_len_extra[sym - 257], &x)) return -1;
            len = skz_len_base[sym - 257] + x;
            ds = skz_real_symbol(s, &dst);
            if (ds < 0 || ds >= 30) return -1;
            if (skz_real_bits(s, skz_dist_extra[ds], &x)) return -1;
            dist = skz_dist_base[ds] + x;
            if (dist > (size_t)(s->out - s->out_base)) return -1;
            for (d = s->out, s->out += len; d < s->out; d++) *d = *(d - dist);
        }
    }
}

/* one compressed block's symbols.  The decoder state lives in local variables for the duration (byte stores
 * through `out` would otherwise force every field of *s to be reloaded). */
static int skz_block(skz_stream *s, const skz_tables *t)
{
    const uint32_t *const lt = t->litlen, *const dt = t->dist;
    const unsigned char *in = s->in, *const in_end = s->in_end;
    uint64_t bitbuf = s->bitbuf;
    unsigned bitcnt = s->bitcnt;
    unsigned char *out = s->out;
    unsigned char *const out_base = s->out_base;            /* (locals: byte stores through `out` may alias every field of *s) */
    const unsigned char *out_lim = s->out_end - (6 + 2 * 258 + 16);
    const unsigned char *const in_fast = in_end - 16;       /* up to here a 64-bit refill stays inside the input */
    int rc;
#define SKZ_SAVE()    do { s->in = in; s->bitbuf = bitbuf; s->bitcnt = bitcnt; s->out = out; } while (0)
#define SKZ_LOAD()    do { out = s->out; } while (0)
#define SKZ_FILL_FAST() do { uint64_t w_; memcpy(&w_, in, 8); bitbuf |= w_ << bitcnt; in += (63u - bitcnt) >> 3; bitcnt |= 56u; } while (0)
#define SKZ_KIND(e)   (((e) >> 4) & 15u)
    uint32_t enext = 0;
    int pre = 0;                                            /* enext = the entry after a match, looked up before its copy */
    for (;;) {
        uint32_t e;
        if (out > out_lim) {
            SKZ_SAVE();
            if (skz_flush(s, 1)) return 1;
            SKZ_LOAD();
        }
        if (pre || in <= in_fast) {
            if (pre) { e = enext; pre = 0; }
            else {
                SKZ_FILL_FAST();
                e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
            }
            /* up to three entries out of one refill (3 x 15 <= 56 bits), each one literal or a pair of them */
#define SKZ_IS_LITS(e) (SKZ_KIND(e) <= (uint32_t)SKZ_K_LIT2)
            if (SKZ_IS_LITS(e)) {
                bitbuf >>= e & 15u; bitcnt -= e & 15u; out[0] = (unsigned char)(e >> 16); out[1] = (unsigned char)(e >> 24); out += 1u + (SKZ_KIND(e) == SKZ_K_LIT2);
                e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
                if (SKZ_IS_LITS(e)) {
                    bitbuf >>= e & 15u; bitcnt -= e & 15u; out[0] = (unsigned char)(e >> 16); out[1] = (unsigned char)(e >> 24); out += 1u + (SKZ_KIND(e) == SKZ_K_LIT2);
                    e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
                    if (SKZ_IS_LITS(e)) {
                        bitbuf >>= e & 15u; bitcnt -= e & 15u; out[0] = (unsigned char)(e >> 16); out[1] = (unsigned char)(e >> 24); out += 1u + (SKZ_KIND(e) == SKZ_K_LIT2);
                        continue;
                    }
                }
                if (in > in_fast + 8) goto tail;            /* (after a preloaded entry the 16-byte margin may be gone) */
                SKZ_FILL_FAST();                            /* (the low bits, hence e, are unchanged) */
            }
        } else {
tail:       /* within 16 bytes of the end of the input: real bits only from here on (e's bits are still in the buffer) */
            SKZ_SAVE();
            return skz_tail(s, t);
        }
        /* here: at least 56 real bits and e is not a literal */
        if (SKZ_KIND(e) == SKZ_K_SUB) {
            bitbuf >>= SKZ_LITLEN_BITS; bitcnt -= SKZ_LITLEN_BITS;
            e = lt[(e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((e >> 8) & 255u)) - 1u))];
            if (SKZ_KIND(e) == SKZ_K_LIT) { bitbuf >>= e & 15u; bitcnt -= e & 15u; *out++ = (unsigned char)(e >> 16); continue; }
        }
        bitbuf >>= e & 15u; bitcnt -= e & 15u;
        if (SKZ_KIND(e) == SKZ_K_LEN) {
            uint32_t len, dist, d;
            const uint32_t xb = (e >> 8) & 255u;
            unsigned char *dst;
            const unsigned char *src;
            len = (e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << xb) - 1u)); bitbuf >>= xb; bitcnt -= xb;   /* <= 20 bits so far */
            d = dt[bitbuf & ((1u << SKZ_DIST_BITS) - 1u)];
            if (SKZ_KIND(d) == SKZ_K_SUB) {
                bitbuf >>= SKZ_DIST_BITS; bitcnt -= SKZ_DIST_BITS;
                d = dt[(d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((d >> 8) & 255u)) - 1u))];
            }
            bitbuf >>= d & 15u; bitcnt -= d & 15u;                               /* <= 35 bits */
            if (SKZ_KIND(d) != SKZ_K_DIST) { rc = -1; break; }
            {
                const uint32_t db = (d >> 8) & 255u;                             /* <= 13: 48 of the 56 bits at most */
                dist = (d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << db) - 1u)); bitbuf >>= db; bitcnt -= db;
            }
            if (dist > (size_t)(out - out_base)) { rc = -1; break; }             /* before the start of the data */
            if (in <= in_fast) {                            /* look the next entry up now: its latency hides behind the copy */
                SKZ_FILL_FAST();
                enext = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
                pre = 1;
            }
            dst = out; src = dst - dist;
            out += len;
            if (dist >= 8) {
                do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < out);
            } else if (dist == 1) {
                memset(dst, *src, len);
            } else {
                do { *dst++ = *src++; } while (dst < out);
            }
            continue;
        }
        if (SKZ_KIND(e) == SKZ_K_EOB) { rc = 0; break; }
        rc = -1;
        break;
    }
    SKZ_SAVE();
    if (rc == 0 && skz_overrun(s)) rc = -1;
    return rc;
#undef SKZ_SAVE
#undef SKZ_LOAD
#undef SKZ_FILL_FAST
#undef SKZ_KIND
#undef SKZ_IS_LITS
}

/* where the next unread bit is, counted from s->data */
static inline uint64_t skz_bit_position(const skz_stream *s) { return (uint64_t)(s->in - s->data) * 8u - s->bitcnt; }

/* continue reading at bit `bit` of s->data */
static inline void skz_seek_bit(skz_stream *s, uint64_t bit)
{
    s->in = s->data + (bit >> 3);
    s->bitbuf = 0; s->bitcnt = 0;
    if (bit & 7u) { SKZ_REFILL(s); SKZ_DROP(s, bit & 7u); }
}

/* one DEFLATE stream (all blocks of a member); 0 at the end of the final block, 1 stopped by the consumer, 2 at
 * s->stop_bit, -1 corrupt */
static int skz_inflate(skz_stream *s, skz_tables *dyn, const skz_tables *fixed)
{
    for (;;) {
        uint32_t final, type;
        int rc, cut = 0;
        if (s->stop_bit && skz_bit_position(s) >= s->stop_bit) return 2;
        SKZ_REFILL(s);
        final = SKZ_BITS(s, 1); SKZ_DROP(s, 1);
        type = SKZ_BITS(s, 2); SKZ_DROP(s, 2);
        if (type == 0) {                                   /* stored: byte-aligned LEN, ~LEN, bytes */
            uint32_t len, nlen;
            SKZ_DROP(s, s->bitcnt & 7u);
            s->in -= s->bitcnt >> 3;                        /* give the whole bytes in the bit buffer back */
            s->bitbuf = 0; s->bitcnt = 0;
            if (s->in + 4 > s->in_end) return -1;
            len = (uint32_t)s->in[0] | ((uint32_t)s->in[1] << 8);
            nlen = (uint32_t)s->in[2] | ((uint32_t)s->in[3] << 8);
            s->in += 4;
            if ((len ^ 0xFFFFu) != nlen) return -1;
            if ((size_t)(s->in_end - s->in) < len) { cut = 1; len = (uint32_t)(s->in_end - s->in); }   /* zlib copies what there is */
            while (len) {
                size_t room = (size_t)(s->out_end - s->out), take;
                if (room < 4096) { if (skz_flush(s, 1)) return 1; room = (size_t)(s->out_end - s->out); }
                take = len < room ? len : room;
                memcpy(s->out, s->in, take);
                s->out += take; s->in += take; len -= (uint32_t)take;
            }
            if (cut) return -1;
        } else if (type == 1) {
            if ((rc = skz_block(s, fixed)) != 0) return rc;
        } else if (type == 2) {
            if (skz_read_dynamic(s, dyn, 0)) return -1;
            if ((rc = skz_block(s, dyn)) != 0) return rc;
        } else return -1;
        if (final) return 0;
    }
}

/* length of the gzip member header at p, 0 if there is none / it is cut short */
static size_t skz_header(const unsigned char *p, size_t n)
{
    size_t i = 10;
    unsigned flg;
    if (n < 18 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || (p[3] & 0xE0)) return 0;
    flg = p[3];
    if (flg & 4) { size_t xlen; if (i + 2 > n) return 0; xlen = (size_t)p[i] | ((size_t)p[i + 1] << 8); i += 2 + xlen; if (i > n) return 0; }
    if (flg & 8) { while (i < n && p[i]) i++; if (i >= n) return 0; i++; }
    if (flg & 16) { while (i < n && p[i]) i++; if (i >= n) return 0; i++; }
    if (flg & 2) i += 2;
    return i <= n ? i : 0;
}

/* where to pick a member up in the middle (sk_gzpar.h hands the rest of a file over this way): the next block
 * starts at bit `bit` of the file, the last `wlen` bytes produced so far are `window`, and crc/total cover
 * everything of this member already delivered */
typedef struct { uint64_t bit; const unsigned char *window; size_t wlen; uint32_t crc; uint64_t total; int members; } skz_resume;

/* members from byte `pos` on (or, with rs, from the middle of the member described there) */
static int skz_decode_members(const unsigned char *data, size_t n, size_t pos, const skz_resume *rs, skz_sink sink, void *user)
{
    skz_stream s;
    skz_tables *dyn, *fixed;
    size_t h;
    int rc = SKZ_OK, members = rs ? rs->members : 0;
    pthread_once(&skz_crc_once, skz_crc_init);
    memset(&s, 0, sizeof s);
    dyn = (skz_tables *)malloc(sizeof *dyn);
    fixed = (skz_tables *)malloc(sizeof *fixed);
    s.out_base = (unsigned char *)malloc(SKZ_WINDOW + SKZ_OUT_CHUNK + 1024);
    if (!dyn || !fixed || !s.out_base) { free(dyn); free(fixed); free(s.out_base); return SKZ_CORRUPT; }
    skz_fixed_tables(fixed);
    s.out_end = s.out_base + SKZ_WINDOW + SKZ_OUT_CHUNK;
    s.sink = sink; s.user = user;
    s.data = data; s.in_end = data + n;
    while (rs || pos < n) {
        int r;
        if (rs) {
            skz_seek_bit(&s, rs->bit);
            memcpy(s.out_base, rs->window, rs->wlen);
            s.out = s.out_flushed = s.out_base + rs->wlen;
            s.crc = rs->crc; s.total = rs->total;
            rs = NULL;
        } else {
            h = skz_header(data + pos, n - pos);
            if (h == 0) { if (!members) rc = SKZ_CORRUPT; break; }       /* trailing bytes after the last member: ignored */
            s.in = data + pos + h;
            s.bitbuf = 0; s.bitcnt = 0;
            s.out = s.out_flushed = s.out_base;                          /* a member cannot refer back into the previous one */
            s.crc = 0; s.total = 0;
        }
        r = skz_inflate(&s, dyn, fixed);
        skz_flush(&s, 0);
        if (r == 1 || s.stopped) { rc = SKZ_STOPPED; break; }
        if (r < 0) { rc = SKZ_CORRUPT; break; }
        /* trailer: whole bytes still in the bit buffer go back to the input first */
        s.in -= s.bitcnt >> 3;
        if (s.in + 8 > s.in_end) { rc = SKZ_CORRUPT; break; }
        {
            const uint32_t crc = (uint32_t)s.in[0] | ((uint32_t)s.in[1] << 8) | ((uint32_t)s.in[2] << 16) | ((uint32_t)s.in[3] << 24);
            const uint32_t isz = (uint32_t)s.in[4] | ((uint32_t)s.in[5] << 8) | ((uint32_t)s.in[6] << 16) | ((uint32_t)s.in[7] << 24);
            if (crc != s.crc || isz != (uint32_t)s.total) { rc = SKZ_CORRUPT; break; }
        }
        pos = (size_t)(s.in + 8 - data);
        members++;
    }
    free(dyn); free(fixed); free(s.out_base);
    return rc;
}

static int skz_decode_memory(const unsigned char *data, size_t n, skz_sink sink, void *user)
{
    if (skz_header(data, n) == 0) return SKZ_NOT_GZIP;
    return skz_decode_members(data, n, 0, NULL, sink, user);
}

__attribute__((unused))
static int skz_decode_file(const char *path, skz_sink sink, void *user)
{
    const int fd = open(path, O_RDONLY);
    struct stat st;
    unsigned char *map;
    int rc;
    if (fd < 0) return SKZ_OPEN;
    if (fstat(fd, &st) || !S_ISREG(st.st_mode) || st.st_size < 18) { close(fd); return SKZ_NOT_GZIP; }
    map = (unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return SKZ_NOT_GZIP;
    madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
    rc = skz_decode_memory(map, (size_t)st.st_size, sink, user);
    munmap(map, (size_t)st.st_size);
    return rc;
}

#endif
