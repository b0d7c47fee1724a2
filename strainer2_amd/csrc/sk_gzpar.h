/* sk_gzpar.h -- one gzip member inflated by several threads.
 *
 * A metagenome usually arrives as one or two big .gz files, and DEFLATE is serial by construction: every block
 * may copy from the 32 KiB before it, and blocks start at arbitrary bit positions that only the decoder of the
 * block before knows.  sk_gzfast.h alone therefore tops out at one core's inflate rate per file (~0.5 GB/s of
 * text) while the record parser behind it does several GB/s and the scan kernel three orders more.
 *
 * What is done here (the idea is the one of pugz / rapidgzip, written from scratch):
 *   - the compressed member is cut into segments of SKZQ_SEG bytes; segment 0 starts where the member starts,
 *     every other segment's thread SEARCHES its first block: the first bit position at or after the segment's
 *     nominal start where a dynamic-Huffman block header parses under zlib's own validity rules and the block
 *     behind it decodes;
 *   - such a segment is decoded SPECULATIVELY into 16-bit symbols: the unknown 32 KiB before it are 32768
 *     placeholder symbols (256 + position), literals are 0..255, and match copies move symbols, placeholders
 *     included.  It stops at the first block boundary at or past the next segment's nominal start;
 *   - the CHAIN step runs in segment order and is cheap: segment k-1 tells where it really ended and what the
 *     last 32 KiB of real bytes are.  If segment k's guessed start is exactly that bit, its symbols are
 *     translated to bytes through a 33024-entry table (in parallel with everything else); if the guess lies
 *     later (the block at the boundary was a stored / fixed / final one the search does not look for) the few
 *     blocks in between are decoded first; anything else (no block found, a false positive, a decode fault)
 *     and the segment is simply decoded again the ordinary way from the known position with the known window;
 *   - the calling thread hands the segments' bytes to the sink in order, computes the member's CRC-32 and
 *     length over them and checks the trailer.
 * Every guess is verified by the chain step (bit-exact hand-over) and the whole by the CRC, so the bytes
 * delivered are those of the serial decoder, whatever the search finds -- tests/test_gzfast.py compares them
 * with zlib's on many kinds of files, with tiny segments so that every path above is taken.
 *
 * Contract: that of skz_decode_memory (same return codes; on a corrupt stream everything decoded before the
 * fault is delivered first).
 */
#ifndef SK_GZPAR_H
#define SK_GZPAR_H
#include "sk_gzfast.h"

#define SKZQ_SEG        (2u << 20)
#define SKZQ_MAX_RATIO  24u                     /* a speculative segment may grow to this many times its compressed size */
#define SKZQ_MAX_THREADS 32

typedef struct {
    uint64_t idx;                               /* the segment this slot holds */
    /* speculative decode */
    int spec_ok, spec_final;
    uint64_t spec_start, spec_end;
    uint16_t *o16; size_t n16, cap16;           /* symbols; the first 32768 are the placeholders (the worker's buffer, lent) */
    int use16;                                  /* the symbols are this segment's output */
    /* set by the chain step */
    int chained, dead, corrupt, final;
    uint64_t end_bit;
    unsigned char window[SKZ_WINDOW]; size_t wlen;      /* last bytes up to this segment's end, right-aligned */
    unsigned char *pre; size_t npre, cappre;    /* bytes decoded the ordinary way (gap or whole segment) */
    unsigned char *o8; size_t n8, cap8;         /* the translated symbols (the buffer stays with the slot) */
    int ready;
} skzq_seg;

typedef struct {
    const unsigned char *data; size_t n;
    uint64_t first_bit;
    size_t base, seg_bytes;
    uint64_t nseg;
    pthread_mutex_t mu; pthread_cond_t cv;
    uint64_t next, consumed;
    int cancel;
    unsigned nslot;
    skzq_seg *slot;
    const skz_tables *fixed;
} skzq_pass;

typedef struct { skz_tables *dyn; unsigned char *outbuf; unsigned char *lut; uint16_t *o16; size_t cap16; } skzq_scratch;   /* per worker */

/* how the segments went (for tools and tests): guessed start taken as it was / after decoding a gap / decoded again */
static uint64_t skzq_stat_direct, skzq_stat_gap, skzq_stat_again;

/* The big buffers come straight from mmap and stay with their worker / slot for the whole member: handing tens of
 * MB back and forth through malloc makes glibc trim and regrow its heap on every segment (measured: 3x slower). */
/* (sizes in whole pages, as the kernel maps them: ThreadSanitizer forgets a range's history only for the bytes munmap names,
 * and the tail of a last page that comes back in a later mapping then reads as a race with a thread long gone) */
#define SKZQ_PAGES(b) (((size_t)(b) + 4095u) & ~(size_t)4095u)
static void *skzq_big_alloc(size_t bytes)
{
    void *q = mmap(NULL, SKZQ_PAGES(bytes), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return q == MAP_FAILED ? NULL : q;
}
static void skzq_big_free(void *q, size_t bytes) { if (q) munmap(q, SKZQ_PAGES(bytes)); }
/* contents kept; NULL (and the old block untouched) when it cannot grow */
static void *skzq_big_grow(void *q, size_t old_bytes, size_t new_bytes)
{
    void *r;
    if (!q) return skzq_big_alloc(new_bytes);
    r = mremap(q, SKZQ_PAGES(old_bytes), SKZQ_PAGES(new_bytes), MREMAP_MAYMOVE);
    return r == MAP_FAILED ? NULL : r;
}

static inline uint64_t skzq_boundary(const skzq_pass *p, uint64_t k) { return ((uint64_t)p->base + k * (uint64_t)p->seg_bytes) * 8u; }
static inline int skzq_cancelled(const skzq_pass *p) { return __atomic_load_n(&p->cancel, __ATOMIC_RELAXED); }

static void skzq_window_push(unsigned char *w, size_t *wlen, const unsigned char *src, size_t n)
{
    if (n >= SKZ_WINDOW) { memcpy(w, src + n - SKZ_WINDOW, SKZ_WINDOW); *wlen = SKZ_WINDOW; return; }
    if (n == 0) return;
    memmove(w, w + n, SKZ_WINDOW - n);
    memcpy(w + SKZ_WINDOW - n, src, n);
    *wlen = *wlen + n > SKZ_WINDOW ? SKZ_WINDOW : *wlen + n;
}

/* ---- the speculative decoder: symbols instead of bytes ------------------------------------------------- */

static int skzq_room16(skzq_seg *g, size_t need, size_t max16)
{
    if (g->n16 + need <= g->cap16) return 0;
    {
        size_t nc = g->cap16 + g->cap16 / 2 + need;
        uint16_t *q;
        if (g->n16 + need > max16) return -1;
        if (nc > max16) nc = max16;
        q = (uint16_t *)skzq_big_grow(g->o16, g->cap16 * sizeof *q, nc * sizeof *q);
        if (!q) return -1;
        g->o16 = q; g->cap16 = nc;
    }
    return 0;
}

/* the symbols of one compressed block (the counterpart of skz_block); 0 at the end-of-block code, -1 on a fault */
static int skzq_block16(skz_stream *s, skzq_seg *g, const skz_tables *t, size_t max16)
{
    const uint32_t *const lt = t->litlen, *const dt = t->dist;
    const unsigned char *in = s->in, *const in_end = s->in_end;
    uint64_t bitbuf = s->bitbuf;
    unsigned bitcnt = s->bitcnt;
    uint16_t *o = g->o16;
    size_t n = g->n16, cap = g->cap16;
    int rc;
    for (;;) {
        uint32_t e, kind;
        if (cap - n < 2 * 258 + 16) {
            g->n16 = n;
            if (skzq_room16(g, 4096, max16)) { rc = -1; break; }
            o = g->o16; cap = g->cap16;
        }
        if (in_end - in >= 8) { uint64_t w_; memcpy(&w_, in, 8); bitbuf |= w_ << bitcnt; in += (63u - bitcnt) >> 3; bitcnt |= 56u; }
        else {
            if (in - in_end > 16) { rc = -1; break; }
            while (bitcnt <= 56u) { const uint64_t b_ = in < in_end ? *in : 0u; in++; bitbuf |= b_ << bitcnt; bitcnt += 8u; }
        }
        /* up to three literal entries per refill, as in skz_block */
        e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
        kind = (e >> 4) & 15u;
        if (kind <= (uint32_t)SKZ_K_LIT2) {
            bitbuf >>= e & 15u; bitcnt -= e & 15u; o[n] = (uint16_t)((e >> 16) & 255u); o[n + 1] = (uint16_t)(e >> 24); n += 1u + (kind == SKZ_K_LIT2);
            e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
            kind = (e >> 4) & 15u;
            if (kind <= (uint32_t)SKZ_K_LIT2) {
                bitbuf >>= e & 15u; bitcnt -= e & 15u; o[n] = (uint16_t)((e >> 16) & 255u); o[n + 1] = (uint16_t)(e >> 24); n += 1u + (kind == SKZ_K_LIT2);
                e = lt[bitbuf & ((1u << SKZ_LITLEN_BITS) - 1u)];
                kind = (e >> 4) & 15u;
                if (kind <= (uint32_t)SKZ_K_LIT2) {
                    bitbuf >>= e & 15u; bitcnt -= e & 15u; o[n] = (uint16_t)((e >> 16) & 255u); o[n + 1] = (uint16_t)(e >> 24); n += 1u + (kind == SKZ_K_LIT2);
                    continue;
                }
            }
            if (bitcnt < 48u) continue;                     /* (a length + distance needs up to 48 bits: refill first) */
        }
        if (kind == SKZ_K_SUB) {
            bitbuf >>= SKZ_LITLEN_BITS; bitcnt -= SKZ_LITLEN_BITS;
            e = lt[(e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((e >> 8) & 255u)) - 1u))];
            kind = (e >> 4) & 15u;
            if (kind == SKZ_K_LIT) { bitbuf >>= e & 15u; bitcnt -= e & 15u; o[n++] = (uint16_t)(e >> 16); continue; }
        }
        bitbuf >>= e & 15u; bitcnt -= e & 15u;
        if (kind == SKZ_K_LEN) {
            const uint32_t xb = (e >> 8) & 255u;
            uint32_t len, dist, d, db;
            len = (e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << xb) - 1u)); bitbuf >>= xb; bitcnt -= xb;
            d = dt[bitbuf & ((1u << SKZ_DIST_BITS) - 1u)];
            if (((d >> 4) & 15u) == SKZ_K_SUB) {
                bitbuf >>= SKZ_DIST_BITS; bitcnt -= SKZ_DIST_BITS;
                d = dt[(d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((d >> 8) & 255u)) - 1u))];
            }
            bitbuf >>= d & 15u; bitcnt -= d & 15u;
            if (((d >> 4) & 15u) != SKZ_K_DIST) { rc = -1; break; }
            db = (d >> 8) & 255u;
            dist = (d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << db) - 1u)); bitbuf >>= db; bitcnt -= db;
            if (dist > n) { rc = -1; break; }               /* (n counts the 32768 placeholders: cannot happen for dist <= 32768) */
            {
                uint16_t *dst = o + n;
                const uint16_t *src = dst - dist;
                n += len;
                if (dist >= 8) {
                    uint16_t *const end = o + n;
                    do { uint64_t w0, w1; memcpy(&w0, src, 8); memcpy(&w1, src + 4, 8); memcpy(dst, &w0, 8); memcpy(dst + 4, &w1, 8); src += 8; dst += 8; } while (dst < end);
                } else {
                    uint16_t *const end = o + n;
                    do { *dst++ = *src++; } while (dst < end);
                }
            }
            continue;
        }
        rc = kind == SKZ_K_EOB ? 0 : -1;
        break;
    }
    s->in = in; s->bitbuf = bitbuf; s->bitcnt = bitcnt;
    g->n16 = n;
    if (rc == 0 && skz_overrun(s)) rc = -1;
    return rc;
}

/* the first bit position in [bit, hi_bit) where a non-final dynamic block header stands up to every check zlib
 * makes; *st is left behind the header with the block's tables in dyn.  UINT64_MAX when there is none. */
static uint64_t skzq_find_block(const skzq_pass *p, uint64_t bit, uint64_t hi_bit, skz_tables *dyn, skz_stream *st)
{
    for (; bit < hi_bit; bit++) {
        const size_t byte = (size_t)(bit >> 3);
        uint64_t w, w2;
        uint32_t hclen, i, sum = 0;
        if (byte + 24 > p->n) break;
        memcpy(&w, p->data + byte, 8);
        w >>= bit & 7u;                                     /* >= 57 bits */
        if ((w & 7u) != 4u) continue;                       /* BFINAL = 0, BTYPE = 2 */
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        hclen = (uint32_t)((w >> 13) & 15u) + 4u;
        memcpy(&w2, p->data + ((bit + 17) >> 3), 8);
        w2 >>= (bit + 17) & 7u;                             /* 19 x 3 = 57 bits */
        for (i = 0; i < hclen; i++) { const uint32_t l = (uint32_t)(w2 >> (3 * i)) & 7u; if (l) sum += 128u >> l; }
        if (sum != 128u) continue;                          /* the code-length code must be complete */
        memset(st, 0, sizeof *st);
        st->data = p->data; st->in_end = p->data + p->n;
        skz_seek_bit(st, bit + 3);
        if (skz_read_dynamic(st, dyn, 1)) continue;
        return bit;
    }
    return UINT64_MAX;
}

static void skzq_speculate(skzq_pass *p, skzq_seg *g, uint64_t k, skzq_scratch *sc)
{
    const uint64_t lo = skzq_boundary(p, k), stop = k + 1 < p->nseg ? skzq_boundary(p, k + 1) : 0;
    const uint64_t hi = stop ? stop : (uint64_t)p->n * 8u;
    const size_t max16 = SKZ_WINDOW + (size_t)SKZQ_MAX_RATIO * p->seg_bytes + 65536;
    uint64_t bit = lo;
    size_t i;
    if (!sc->o16) {
        sc->cap16 = SKZ_WINDOW + 6 * p->seg_bytes + 4096;
        if (sc->cap16 > max16) sc->cap16 = max16;
        sc->o16 = (uint16_t *)skzq_big_alloc(sc->cap16 * sizeof *sc->o16);
        if (!sc->o16) { sc->cap16 = 0; return; }
    }
    g->o16 = sc->o16; g->cap16 = sc->cap16;             /* (handed back, possibly grown, by the worker loop) */
    for (i = 0; i < SKZ_WINDOW; i++) g->o16[i] = (uint16_t)(256u + i);
    for (;;) {
        skz_stream st;
        int rc;
        bit = skzq_find_block(p, bit, hi, sc->dyn, &st);
        if (bit == UINT64_MAX) return;
        g->n16 = SKZ_WINDOW;
        g->spec_start = bit;
        rc = skzq_block16(&st, g, sc->dyn, max16);
        if (rc) { bit++; continue; }                        /* a header that only looked like one */
        for (;;) {
            uint32_t final, type;
            const uint64_t at = skz_bit_position(&st);
            if (skzq_cancelled(p)) return;
            if (stop && at >= stop) { g->spec_end = at; g->spec_final = 0; g->spec_ok = 1; return; }
            SKZ_REFILL(&st);
            final = SKZ_BITS(&st, 1); SKZ_DROP(&st, 1);
            type = SKZ_BITS(&st, 2); SKZ_DROP(&st, 2);
            if (type == 0) {
                uint32_t len, nlen, j;
                SKZ_DROP(&st, st.bitcnt & 7u);
                st.in -= st.bitcnt >> 3;
                st.bitbuf = 0; st.bitcnt = 0;
                if (st.in + 4 > st.in_end) return;
                len = (uint32_t)st.in[0] | ((uint32_t)st.in[1] << 8);
                nlen = (uint32_t)st.in[2] | ((uint32_t)st.in[3] << 8);
                st.in += 4;
                if ((len ^ 0xFFFFu) != nlen || (size_t)(st.in_end - st.in) < len) return;
                if (skzq_room16(g, len, max16)) return;
                for (j = 0; j < len; j++) g->o16[g->n16 + j] = st.in[j];
                g->n16 += len; st.in += len;
            } else if (type == 1) {
                if (skzq_block16(&st, g, p->fixed, max16)) return;
            } else if (type == 2) {
                if (skz_read_dynamic(&st, sc->dyn, 0) || skzq_block16(&st, g, sc->dyn, max16)) return;
            } else return;
            if (final) { g->spec_end = skz_bit_position(&st); g->spec_final = 1; g->spec_ok = 1; return; }
        }
    }
}

/* ---- the ordinary decoder over a bit range, into g->pre ------------------------------------------------- */

static int skzq_pre_sink(void *user, const unsigned char *d, size_t n)
{
    skzq_seg *g = (skzq_seg *)user;
    if (g->npre + n > g->cappre) {
        const size_t nc = (g->npre + n) + (g->npre + n) / 2 + (1u << 20);
        unsigned char *q = (unsigned char *)skzq_big_grow(g->pre, g->cappre, nc);
        if (!q) return 1;
        g->pre = q; g->cappre = nc;
    }
    memcpy(g->pre + g->npre, d, n);
    g->npre += n;
    return 0;
}

/* blocks from bit `from` (history: the wlen bytes right-aligned in window) up to the first boundary at or past
 * `stop` (0: to the end of the member).  Returns 0 at a boundary, 1 after the final block, -1 corrupt (what was
 * decoded before the fault is in g->pre) */
static int skzq_serial(skzq_pass *p, skzq_scratch *sc, skzq_seg *g, uint64_t from, const unsigned char *window, size_t wlen,
                       uint64_t stop, uint64_t *end_bit)
{
    skz_stream s;
    int r;
    memset(&s, 0, sizeof s);
    s.data = p->data; s.in_end = p->data + p->n;
    s.out_base = sc->outbuf; s.out_end = sc->outbuf + SKZ_WINDOW + SKZ_OUT_CHUNK;
    s.sink = skzq_pre_sink; s.user = g; s.skip_crc = 1; s.stop_bit = stop;
    skz_seek_bit(&s, from);
    memcpy(s.out_base, window + SKZ_WINDOW - wlen, wlen);
    s.out = s.out_flushed = s.out_base + wlen;
    r = skz_inflate(&s, sc->dyn, p->fixed);
    skz_flush(&s, 0);
    *end_bit = skz_bit_position(&s);
    if (s.stopped) return -1;                               /* (out of memory in the sink) */
    return r == 0 ? 1 : r == 2 ? 0 : -1;
}

/* symbols to bytes: 16 at a time where none of them is a placeholder (most of a FASTQ file is not) */
static void skzq_translate(unsigned char *dst, const uint16_t *src, size_t n, const unsigned char *lut)
{
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i top = _mm_set1_epi16((short)0xFF00);
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 8));
        if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(_mm_or_si128(a, b), top), _mm_setzero_si128())) == 0xFFFF)
            _mm_storeu_si128((__m128i *)(dst + i), _mm_packus_epi16(a, b));
        else {
            size_t j;
            for (j = i; j < i + 16; j++) dst[j] = lut[src[j]];
        }
    }
#endif
    for (; i < n; i++) dst[i] = lut[src[i]];
}

/* ---- chain step: make segment k's result definite, given where k-1 ended -------------------------------- */

static void skzq_chain(skzq_pass *p, skzq_scratch *sc, skzq_seg *g, uint64_t k, uint64_t prev_end,
                       const unsigned char *pwin, size_t pwlen)
{
    const uint64_t stop = k + 1 < p->nseg ? skzq_boundary(p, k + 1) : 0;
    int use = 0, gap = 0;
    memcpy(g->window, pwin, SKZ_WINDOW);
    g->wlen = pwlen;
    if (g->spec_ok && g->spec_start >= prev_end) {
        use = 1;
        if (g->spec_start > prev_end) {                     /* blocks the search skipped: decode up to the guess */
            uint64_t e = 0;
            const int r = skzq_serial(p, sc, g, prev_end, pwin, pwlen, g->spec_start, &e);
            gap = 1;
            if (r != 0 || e != g->spec_start) { use = 0; g->npre = 0; }
            else skzq_window_push(g->window, &g->wlen, g->pre, g->npre);
        }
        if (use && g->wlen < SKZ_WINDOW) {                  /* less than 32 KiB of history: every placeholder must lie inside it */
            const uint16_t lowest = (uint16_t)(256u + (SKZ_WINDOW - g->wlen));
            size_t i;
            for (i = SKZ_WINDOW; i < g->n16; i++) if (g->o16[i] >= 256u && g->o16[i] < lowest) { use = 0; break; }
            if (!use) { g->npre = 0; memcpy(g->window, pwin, SKZ_WINDOW); g->wlen = pwlen; }
        }
    }
    __atomic_fetch_add(use ? (gap ? &skzq_stat_gap : &skzq_stat_direct) : &skzq_stat_again, 1, __ATOMIC_RELAXED);
    g->use16 = use;
    if (use) {
        const size_t nout = g->n16 - SKZ_WINDOW;
        size_t i;
        for (i = 0; i < 256; i++) sc->lut[i] = (unsigned char)i;
        memcpy(sc->lut + 256, g->window, SKZ_WINDOW);
        g->end_bit = g->spec_end; g->final = g->spec_final;
        /* the window after this segment, so that the next one can go on at once */
        if (nout >= SKZ_WINDOW) {
            for (i = 0; i < SKZ_WINDOW; i++) g->window[i] = sc->lut[g->o16[g->n16 - SKZ_WINDOW + i]];
            g->wlen = SKZ_WINDOW;
        } else {
            unsigned char tail[SKZ_WINDOW];
            for (i = 0; i < nout; i++) tail[i] = sc->lut[g->o16[SKZ_WINDOW + i]];
            skzq_window_push(g->window, &g->wlen, tail, nout);
        }
    } else {
        uint64_t e = 0;
        int r;
        r = skzq_serial(p, sc, g, prev_end, pwin, pwlen, stop, &e);
        g->end_bit = e;
        g->final = r == 1;
        g->corrupt = r < 0;
        skzq_window_push(g->window, &g->wlen, g->pre, g->npre);
    }
}

static void *skzq_worker(void *arg)
{
    skzq_pass *p = (skzq_pass *)arg;
    skzq_scratch sc;
    sc.dyn = (skz_tables *)malloc(sizeof *sc.dyn);
    sc.outbuf = (unsigned char *)skzq_big_alloc(SKZ_WINDOW + SKZ_OUT_CHUNK + 4096);
    sc.lut = (unsigned char *)malloc(256 + SKZ_WINDOW);
    sc.o16 = NULL; sc.cap16 = 0;
    for (;;) {
        skzq_seg *g, *prev = NULL;
        uint64_t k, prev_end;
        int prev_over = 0, oom = !sc.dyn || !sc.outbuf || !sc.lut;
        pthread_mutex_lock(&p->mu);
        while (!p->cancel && p->next < p->nseg && p->next + 1 >= p->consumed + p->nslot) pthread_cond_wait(&p->cv, &p->mu);
        if (p->cancel || p->next >= p->nseg) { pthread_mutex_unlock(&p->mu); break; }
        k = p->next++;
        g = &p->slot[k % p->nslot];
        g->idx = k;
        g->spec_ok = g->spec_final = g->chained = g->dead = g->corrupt = g->final = g->ready = 0;
        g->n16 = g->cap16 = g->npre = g->n8 = 0;           /* (the consumer freed the buffers) */
        g->o16 = NULL; g->use16 = 0;
        pthread_mutex_unlock(&p->mu);
        if (k > 0 && !oom) {
            skzq_speculate(p, g, k, &sc);
            if (g->o16) { sc.o16 = g->o16; sc.cap16 = g->cap16; }      /* (it may have been reallocated) */
        }
        if (k == 0) prev_end = p->first_bit;
        else {
            prev = &p->slot[(k - 1) % p->nslot];
            pthread_mutex_lock(&p->mu);
            while (!p->cancel && !(prev->idx == k - 1 && prev->chained)) pthread_cond_wait(&p->cv, &p->mu);
            prev_over = p->cancel || prev->dead || prev->final || prev->corrupt;
            prev_end = prev_over ? 0 : prev->end_bit;      /* (after a cancel, segment k-1 may still be at work) */
            pthread_mutex_unlock(&p->mu);
        }
        if (prev_over || oom) {
            g->o16 = NULL;
            pthread_mutex_lock(&p->mu);
            g->dead = 1; g->corrupt = oom; g->chained = g->ready = 1;
            pthread_cond_broadcast(&p->cv);
            pthread_mutex_unlock(&p->mu);
            continue;
        }
        {   /* (segment k-1's slot is not reused before segment k has been delivered) */
            static const unsigned char none[SKZ_WINDOW] = {0};
            skzq_chain(p, &sc, g, k, prev_end, prev ? prev->window : none, prev ? prev->wlen : 0);
        }
        pthread_mutex_lock(&p->mu);
        g->chained = 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
        if (g->use16) {
            const size_t nout = g->n16 - SKZ_WINDOW;
            if (nout + 16 > g->cap8) {
                skzq_big_free(g->o8, g->cap8);
                g->cap8 = nout + nout / 8 + (1u << 20);
                g->o8 = (unsigned char *)skzq_big_alloc(g->cap8);
                if (!g->o8) g->cap8 = 0;
            }
            if (g->o8) { skzq_translate(g->o8, g->o16 + SKZ_WINDOW, nout, sc.lut); g->n8 = nout; }
            else g->corrupt = 1;
        }
        g->o16 = NULL;
        pthread_mutex_lock(&p->mu);
        g->ready = 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
    }
    free(sc.dyn); free(sc.lut);
    skzq_big_free(sc.outbuf, SKZ_WINDOW + SKZ_OUT_CHUNK + 4096);
    skzq_big_free(sc.o16, sc.cap16 * sizeof *sc.o16);
    return NULL;
}

/* one member whose DEFLATE data starts at byte `start`: SKZ_OK with *next = the byte after its trailer and
 * *short_member set when it ended within the first two segments; SKZ_CORRUPT / SKZ_STOPPED otherwise */
static int skzq_member(const unsigned char *data, size_t n, size_t start, int nthreads, size_t seg_bytes, const skz_tables *fixed,
                       skz_sink sink, void *user, size_t *next, int *short_member)
{
    skzq_pass p;
    pthread_t th[SKZQ_MAX_THREADS];
    int nth = 0, i, rc = SKZ_CORRUPT;
    uint32_t crc = 0;
    uint64_t total = 0, k;
    memset(&p, 0, sizeof p);
    p.data = data; p.n = n; p.base = start; p.first_bit = (uint64_t)start * 8u; p.seg_bytes = seg_bytes; p.fixed = fixed;
    p.nseg = ((uint64_t)(n - start) + seg_bytes - 1) / seg_bytes;
    if (nthreads > SKZQ_MAX_THREADS) nthreads = SKZQ_MAX_THREADS;
    p.nslot = (unsigned)nthreads + 4u;
    p.slot = (skzq_seg *)calloc(p.nslot, sizeof *p.slot);
    if (!p.slot) return SKZ_CORRUPT;
    for (i = 0; i < (int)p.nslot; i++) p.slot[i].idx = UINT64_MAX;
    pthread_mutex_init(&p.mu, NULL);
    pthread_cond_init(&p.cv, NULL);
    for (i = 0; i < nthreads; i++) if (pthread_create(&th[nth], NULL, skzq_worker, &p) == 0) nth++;
    if (nth == 0) { rc = -100; goto out; }                  /* no thread at all: the caller decodes serially */
    for (k = 0; k < p.nseg; k++) {
        skzq_seg *g = &p.slot[k % p.nslot];
        int stop = 0;
        pthread_mutex_lock(&p.mu);
        while (!(g->idx == k && g->ready)) pthread_cond_wait(&p.cv, &p.mu);
        pthread_mutex_unlock(&p.mu);
        if (g->dead && !g->corrupt) break;                  /* (cannot happen before a final or corrupt segment) */
        if (g->npre) { crc = skz_crc32(crc, g->pre, g->npre); total += g->npre; if (sink(user, g->pre, g->npre)) stop = 1; }
        if (g->n8 && !stop) { crc = skz_crc32(crc, g->o8, g->n8); total += g->n8; if (sink(user, g->o8, g->n8)) stop = 1; }
        g->n8 = 0; g->npre = 0;
        if (stop) { rc = SKZ_STOPPED; break; }
        if (g->corrupt) break;
        if (g->final) {
            const size_t t = (size_t)((g->end_bit + 7u) >> 3);
            if (t + 8 <= n) {
                const uint32_t c = (uint32_t)data[t] | ((uint32_t)data[t + 1] << 8) | ((uint32_t)data[t + 2] << 16) | ((uint32_t)data[t + 3] << 24);
                const uint32_t z = (uint32_t)data[t + 4] | ((uint32_t)data[t + 5] << 8) | ((uint32_t)data[t + 6] << 16) | ((uint32_t)data[t + 7] << 24);
                if (c == crc && z == (uint32_t)total) { rc = SKZ_OK; *next = t + 8; *short_member = k < 2; }
            }
            break;
        }
        pthread_mutex_lock(&p.mu);
        p.consumed = k + 1;
        pthread_cond_broadcast(&p.cv);
        pthread_mutex_unlock(&p.mu);
    }
out:
    pthread_mutex_lock(&p.mu);
    __atomic_store_n(&p.cancel, 1, __ATOMIC_RELAXED);
    pthread_cond_broadcast(&p.cv);
    pthread_mutex_unlock(&p.mu);
    for (i = 0; i < nth; i++) pthread_join(th[i], NULL);
    for (i = 0; i < (int)p.nslot; i++) { skzq_big_free(p.slot[i].pre, p.slot[i].cappre); skzq_big_free(p.slot[i].o8, p.slot[i].cap8); }
    free(p.slot);
    pthread_mutex_destroy(&p.mu);
    pthread_cond_destroy(&p.cv);
    return rc;
}

/* skz_decode_memory with `nthreads` inflating threads per member (seg_bytes 0: SKZQ_SEG).  Members shorter than a
 * few segments -- and everything after the first such member: a file of many small members -- go the serial way. */
static int skzq_decode_memory(const unsigned char *data, size_t n, int nthreads, size_t seg_bytes, skz_sink sink, void *user)
{
    skz_tables *fixed;
    size_t pos = 0;
    int members = 0, rc = SKZ_OK;
    if (skz_header(data, n) == 0) return SKZ_NOT_GZIP;
    if (nthreads < 2) return skz_decode_members(data, n, 0, NULL, sink, user);
    if (!seg_bytes) seg_bytes = SKZQ_SEG;
    if (seg_bytes < 64) seg_bytes = 64;
    pthread_once(&skz_crc_once, skz_crc_init);
    fixed = (skz_tables *)malloc(sizeof *fixed);
    if (!fixed) return SKZ_CORRUPT;
    skz_fixed_tables(fixed);
    while (pos < n) {
        const size_t h = skz_header(data + pos, n - pos);
        size_t next = 0;
        int short_member = 0;
        if (h == 0) { if (!members) rc = SKZ_CORRUPT; break; }
        if (n - pos - h < 4 * seg_bytes) { rc = skz_decode_members(data, n, pos, NULL, sink, user); break; }
        rc = skzq_member(data, n, pos + h, nthreads, seg_bytes, fixed, sink, user, &next, &short_member);
        if (rc == -100) { rc = skz_decode_members(data, n, pos, NULL, sink, user); break; }
        if (rc != SKZ_OK) break;
        pos = next;
        members++;
        if (short_member && pos < n) {
            if (skz_header(data + pos, n - pos)) rc = skz_decode_members(data, n, pos, NULL, sink, user);
            break;
        }
    }
    free(fixed);
    return rc;
}

#endif
