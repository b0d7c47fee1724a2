/* sk_pyfmt.h -- the text conventions of the two Python consumers this library also replaces
 * (reference scripts/kmer_scrub_filter.py, scripts/coverage_depth.py): how they print floats and read
 * integers.  Host only. */
#ifndef SK_PYFMT_H
#define SK_PYFMT_H
#include <ctype.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* str(float): the shortest digit string that reads back as the same double; positional between 1e-4
 * and 1e16, exponent form outside.  `out` needs 32 bytes. */
static inline void skp_float_str(double x, char *out)
{
    char e[40], dig[24];
    int prec, nd = 0, neg, x10, pt;
    const char *p;
    char *o = out;
    if (x != x) { strcpy(out, "nan"); return; }
    if (x - x != 0.0) { strcpy(out, x > 0 ? "inf" : "-inf"); return; }
    for (prec = 0; prec < 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec, x);
        if (strtod(e, NULL) == x) break;
    }
    neg = e[0] == '-';
    for (p = e + neg; *p != 'e'; p++)
        if (*p != '.') dig[nd++] = *p;
    x10 = atoi(p + 1);
    while (nd > 1 && dig[nd - 1] == '0') nd--;
    pt = x10 + 1;                                  /* decimal point sits after dig[pt-1] */
    if (neg) *o++ = '-';
    if (pt > 16 || pt < -3) {
        *o++ = dig[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, dig + 1, (size_t)nd - 1); o += nd - 1; }
        sprintf(o, "e%c%02d", x10 < 0 ? '-' : '+', x10 < 0 ? -x10 : x10);
        return;
    }
    if (pt <= 0) {
        *o++ = '0'; *o++ = '.';
        memset(o, '0', (size_t)-pt); o += -pt;
        memcpy(o, dig, (size_t)nd); o += nd;
    } else if (pt >= nd) {
        memcpy(o, dig, (size_t)nd); o += nd;
        memset(o, '0', (size_t)(pt - nd)); o += pt - nd;
        *o++ = '.'; *o++ = '0';
    } else {
        memcpy(o, dig, (size_t)pt); o += pt;
        *o++ = '.';
        memcpy(o, dig + pt, (size_t)(nd - pt)); o += nd - pt;
    }
    *o = 0;
}

/* int(text) for the decimal fields of the tables: blanks, a sign, digits, blanks.  1 on success. */
static inline int skp_int(const char *s, const char *end, int64_t *out)
{
    int neg = 0;
    uint64_t v = 0;
    const char *d0;
    while (s < end && isspace((unsigned char)*s)) s++;
    while (end > s && isspace((unsigned char)end[-1])) end--;
    if (s < end && (*s == '-' || *s == '+')) neg = *s++ == '-';
    d0 = s;
    for (; s < end && *s >= '0' && *s <= '9'; s++) {
        if (v > (UINT64_MAX - 9) / 10) return 0;
        v = v * 10 + (uint64_t)(*s - '0');
    }
    if (s == d0 || s != end || v > (uint64_t)INT64_MAX) return 0;
    *out = neg ? -(int64_t)v : (int64_t)v;
    return 1;
}

#endif
