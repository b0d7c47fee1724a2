/* sk_parser.h -- FASTA/FASTQ record parser shared by the host-layer sources (internal).
 *
 * Push-style state machine over inflated blocks with the observable behaviour of the reference's
 * parser (src/kseq.h:166-211): what counts as a header, how sequence lines are joined, the CR rule,
 * FASTQ quality accounting, and the three ways a file can end:
 *   END_RESET  (-1 after the lengths were reset: the file ended inside/after a FASTA-style record)
 *   END_STALE  (-1 before the reset: the file ended while seeking the next header after a FASTQ record;
 *               the reference's seq.l still holds the previous record's length)
 *   END_TRUNC  (-2: quality string length mismatch; seq.l is that record's sequence length)
 */
#ifndef SK_PARSER_H
#define SK_PARSER_H
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { SKP_END_NONE = 0, SKP_END_RESET = 1, SKP_END_STALE = 2, SKP_END_TRUNC = 3 };

enum { P_SEEK, P_NAME, P_COMMENT, P_LINE_START, P_SEQ, P_PLUS, P_QUAL, P_STOP };

typedef int (*rec_fn)(void *user, char *seq, size_t len);

typedef struct {
    int     state;
    int     name_any, line_any, qual_any;
    char   *seq;  size_t seq_len, seq_cap;
    char   *qual; size_t qual_len, qual_cap;
    rec_fn  on_record;
    void   *user;
    int64_t nrecords;
    int     sink_rc;
    int     end_kind;       /* SKP_END_* once the input is exhausted or the parser stopped */
    size_t  end_len;        /* the reference's seq.l at that point */
    size_t  last_len;       /* length of the last record handed out */
} parser;

static void grow(char **p, size_t *cap, size_t need)
{
    if (need + 1 > *cap) {
        size_t nc = *cap ? *cap : 4096;
        while (nc < need + 1) nc *= 2;
        *p = (char *)realloc(*p, nc);
        *cap = nc;
    }
}

static void parser_init(parser *ps, rec_fn fn, void *user)
{
    memset(ps, 0, sizeof *ps);
    ps->state = P_SEEK;
    ps->on_record = fn;
    ps->user = user;
}

static void parser_free(parser *ps) { free(ps->seq); free(ps->qual); }

static void emit(parser *ps)
{
    grow(&ps->seq, &ps->seq_cap, ps->seq_len);
    ps->seq[ps->seq_len] = '\0';
    ps->nrecords++;
    ps->last_len = ps->seq_len;
    if (ps->on_record) {
        int rc = ps->on_record(ps->user, ps->seq, ps->seq_len);
        if (rc) { ps->sink_rc = rc; ps->state = P_STOP; }
    }
}

static void begin_header(parser *ps)
{
    ps->state = P_NAME;
    ps->name_any = 0;
    ps->seq_len = 0;
    ps->qual_len = 0;
}

/* a trailing CR is dropped once the accumulated text is longer than one byte (src/kseq.h:136) */
static void strip_cr(char *p, size_t *len) { if (*len > 1 && p[*len - 1] == '\r') (*len)--; }

static void finish_fastq(parser *ps)
{
    if (ps->qual_len == ps->seq_len) { emit(ps); if (ps->state != P_STOP) ps->state = P_SEEK; }
    else { ps->state = P_STOP; ps->end_kind = SKP_END_TRUNC; ps->end_len = ps->seq_len; }   /* -2: dropped, file over */
}

/* The common cases in one step, straight from the block at hand (no copies of the sequence and quality text, no trips through
 * the state machine):
 *   FASTQ  a record whose four lines -- header, ONE sequence line, '+' line, quality line of the same length -- all lie inside
 *          the block;
 *   FASTA  (round 4: a 32-strain pass over a plain FASTA metagenome spent 28 of its 49 CPU-seconds in the general path below --
 *          two copies of every read) a header line and ONE sequence line, followed IN THE BLOCK by the next record's header
 *          character -- which is what tells the state machine, too, that the record is complete (P_LINE_START).
 * Exactly what the states below would do with those bytes, including the CR rule; anything else (wrapped lines, a record cut
 * by the block's end or followed by nothing, an empty or '+'-led sequence line, a quality string of another length) returns
 * 0 untouched and goes the general way.  *at is the header's first byte; on success it is the first byte behind the record
 * and the state stays P_SEEK (for FASTA: the next header's character, found at once). */
static int parser_whole_record(parser *ps, const unsigned char *b, size_t n, size_t *at)
{
    const unsigned char *const end = b + n, *s, *q, *e1, *e2, *e3, *e4;
    size_t len, qlen;
    if (!(e1 = (const unsigned char *)memchr(b + *at + 1, '\n', (size_t)(end - (b + *at + 1))))) return 0;
    s = e1 + 1;
    if (s >= end || *s == '\n' || *s == '>' || *s == '@' || *s == '+') return 0;
    if (!(e2 = (const unsigned char *)memchr(s, '\n', (size_t)(end - s))) || e2 + 1 >= end) return 0;
    len = (size_t)(e2 - s);
    if (len > 1 && s[len - 1] == '\r') len--;
    if (e2[1] == '>' || e2[1] == '@') {                  /* FASTA, one line: the next header ends it */
        ps->name_any = 1; ps->seq_len = len; ps->qual_len = 0;
        ps->nrecords++;
        ps->last_len = len;
        *at = (size_t)(e2 + 1 - b);
    } else {
        if (e2[1] != '+') return 0;
        if (!(e3 = (const unsigned char *)memchr(e2 + 1, '\n', (size_t)(end - (e2 + 1))))) return 0;
        q = e3 + 1;
        if (!(e4 = (const unsigned char *)memchr(q, '\n', (size_t)(end - q)))) return 0;
        qlen = (size_t)(e4 - q);
        if (qlen > 1 && q[qlen - 1] == '\r') qlen--;
        if (qlen != len) return 0;
        ps->name_any = 1; ps->seq_len = len; ps->qual_len = qlen;
        ps->nrecords++;
        ps->last_len = len;
        *at = (size_t)(e4 + 1 - b);
    }
    if (ps->on_record) {
        const int rc = ps->on_record(ps->user, (char *)s, len);   /* (the callbacks read len bytes, none writes) */
        if (rc) { ps->sink_rc = rc; ps->state = P_STOP; }
    }
    return 1;
}

static void parser_feed(parser *ps, const unsigned char *b, size_t n)
{
    size_t i = 0;
    while (i < n && ps->state != P_STOP) {
        switch (ps->state) {
        case P_SEEK:
            while (i < n && b[i] != '>' && b[i] != '@') i++;
            if (i < n && !parser_whole_record(ps, b, n, &i)) { i++; begin_header(ps); }
            break;
        case P_NAME: {
            /* the name ends at the first white space and the rest of the header line is skipped (P_COMMENT), so all
             * that matters is where the line ends; a header cut off by the end of the input gives an empty record
             * from either state (parser_eof) */
            const unsigned char *nl = (const unsigned char *)memchr(b + i, '\n', n - i);
            if (i < n) ps->name_any = 1;
            if (nl) { i = (size_t)(nl - b) + 1; ps->state = P_LINE_START; } else i = n;
            break; }
        case P_COMMENT: {
            const unsigned char *nl = (const unsigned char *)memchr(b + i, '\n', n - i);
            if (nl) { i = (size_t)(nl - b) + 1; ps->state = P_LINE_START; } else i = n;
            break; }
        case P_LINE_START: {
            int c = b[i++];
            if (c == '\n') break;
            if (c == '>' || c == '@') { emit(ps); if (ps->state != P_STOP) begin_header(ps); break; }
            if (c == '+') { ps->state = P_PLUS; break; }
            grow(&ps->seq, &ps->seq_cap, ps->seq_len + 1);
            ps->seq[ps->seq_len++] = (char)c;
            ps->line_any = 0;
            ps->state = P_SEQ;
            break; }
        case P_SEQ: {
            const unsigned char *nl = (const unsigned char *)memchr(b + i, '\n', n - i);
            size_t take = nl ? (size_t)(nl - (b + i)) : n - i;
            ps->line_any = 1;
            grow(&ps->seq, &ps->seq_cap, ps->seq_len + take);
            memcpy(ps->seq + ps->seq_len, b + i, take);
            ps->seq_len += take;
            i += take;
            if (nl) { i++; strip_cr(ps->seq, &ps->seq_len); ps->state = P_LINE_START; }
            break; }
        case P_PLUS: {
            const unsigned char *nl = (const unsigned char *)memchr(b + i, '\n', n - i);
            if (nl) { i = (size_t)(nl - b) + 1; ps->state = P_QUAL; ps->qual_len = 0; ps->qual_any = 0; } else i = n;
            break; }
        case P_QUAL: {
            const unsigned char *nl = (const unsigned char *)memchr(b + i, '\n', n - i);
            size_t take = nl ? (size_t)(nl - (b + i)) : n - i;
            ps->qual_any = 1;
            grow(&ps->qual, &ps->qual_cap, ps->qual_len + take);
            memcpy(ps->qual + ps->qual_len, b + i, take);
            ps->qual_len += take;
            i += take;
            if (nl) {
                i++;
                strip_cr(ps->qual, &ps->qual_len);
                ps->qual_any = 0;
                if (ps->qual_len >= ps->seq_len) finish_fastq(ps);
            }
            break; }
        default: i = n; break;
        }
    }
}

static void parser_eof(parser *ps)
{
    const int st = ps->state;
    if (st == P_STOP) return;
    switch (ps->state) {
    case P_NAME:       if (ps->name_any) emit(ps); break;     /* header cut by EOF: empty record */
    case P_COMMENT:    emit(ps); break;
    case P_LINE_START: emit(ps); break;
    case P_SEQ:        if (ps->line_any) strip_cr(ps->seq, &ps->seq_len); emit(ps); break;
    case P_QUAL:       if (ps->qual_any) strip_cr(ps->qual, &ps->qual_len); finish_fastq(ps); break;
    default: break;                                           /* P_SEEK, P_PLUS (-2) */
    }
    if (ps->end_kind == SKP_END_NONE) {
        if (st == P_SEEK) { ps->end_kind = SKP_END_STALE; ps->end_len = ps->last_len; }
        else if (st == P_PLUS) { ps->end_kind = SKP_END_TRUNC; ps->end_len = ps->seq_len; }
        else { ps->end_kind = SKP_END_RESET; ps->end_len = 0; }
    }
    ps->state = P_STOP;
}


/* ---- cutting a text into pieces that are parsed on their own (several threads, several ranks) -------------------- */
/* First offset >= x of a plain FASTA/FASTQ text at which a record may start: a '>' or '@' at the start of a line --
 * for '@' with a '+' line two lines on, which sets it apart from a quality line that happens to begin with '@'
 * (src/kseq.h:171-211 has no such check, it reads from the top; a guess that is wrong is caught by parse_range). */
static __attribute__((unused)) uint64_t parser_guess_start(const unsigned char *t, uint64_t size, uint64_t x, int size_is_eof)
{
    uint64_t i = x;
    if (x == 0) return 0;
    while (i < size) {
        const unsigned char *nl = (const unsigned char *)memchr(t + i - 1, '\n', (size_t)(size - (i - 1)));
        uint64_t s;
        if (!nl) return size;
        s = (uint64_t)(nl - t) + 1;
        if (s >= size) return size;
        if (t[s] == '>') {
            /* not if the line before begins with '+': then this is a FASTQ quality line that happens to begin with '>'
             * (no line of a FASTA file begins with '+') */
            uint64_t q = s >= 2 ? s - 2 : 0;
            while (q > 0 && t[q] != '\n') q--;
            if (!(s >= 2 && t[q == 0 && t[0] != '\n' ? 0 : q + 1] == '+')) return s;
        }
        if (t[s] == '@') {
            const unsigned char *l1 = (const unsigned char *)memchr(t + s, '\n', (size_t)(size - s));
            const unsigned char *l2 = l1 && (uint64_t)(l1 - t) + 1 < size ? (const unsigned char *)memchr(l1 + 1, '\n', (size_t)(size - ((uint64_t)(l1 - t) + 1))) : NULL;
            if (!l2 || (uint64_t)(l2 - t) + 1 >= size) {                   /* too close to the end of the text at hand to tell */
                if (size_is_eof) return s;                                  /* (the end of the file: the check decides) */
                return size;                                                /* (more text will come: no guess yet) */
            }
            if (l2[1] == '+') return s;
        }
        i = s + 1;
    }
    return size;
}


/* After the bytes of a piece have been fed: does the parser stand BETWEEN two records (after a complete FASTQ record, or at
 * the start of a line after a FASTA header/sequence line), so that a header character next begins a new record?  If the
 * piece began at a true record boundary and this holds at its end, the end is a true boundary as well. */
static __attribute__((unused)) int parser_between_records(const parser *ps) { return ps->state == P_SEEK || ps->state == P_LINE_START; }

#endif
