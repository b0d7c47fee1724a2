/* tools/probes/parse_probe.c -- what ONE parser thread delivers: a plain FASTA/FASTQ file, mapped, fed to the product's record
 * parser (strainer2_amd/csrc/sk_parser.h) in blocks, every record of k bases or more copied behind the others with a '\n' as the
 * chunk builders do.  Prints GB/s of file and Gbase/s.   gcc -O2 -o tools/probes/parse_probe tools/probes/parse_probe.c
 *   [OUT_MB=640] tools/probes/parse_probe FILE [block bytes, default 4 MiB] [HEADER=path of another sk_parser.h to compare: compile time only] */
#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifndef PARSER_H
#define PARSER_H "../../strainer2_amd/csrc/sk_parser.h"
#endif
#include PARSER_H

static unsigned char *out; static size_t out_len, out_cap; static unsigned long long nrec, nbases;
static int on_rec(void *u, char *seq, size_t len)
{
    (void)u;
    nrec++;
    if (len < 31) return 0;
    if (out_len + len + 1 > out_cap) out_len = 0;        /* (a chunk is full: the next one) */
    memcpy(out + out_len, seq, len); out[out_len + len] = '\n'; out_len += len + 1;
    nbases += len;
    return 0;
}
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char **argv)
{
    size_t blk = argc > 2 ? (size_t)atol(argv[2]) : 4u << 20, at;
    struct stat sb;
    int fd = open(argv[1], O_RDONLY), rep;
    const unsigned char *m;
    if (fd < 0 || fstat(fd, &sb)) { perror(argv[1]); return 1; }
    m = (const unsigned char *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    out_cap = (size_t)(getenv("OUT_MB") ? atol(getenv("OUT_MB")) : 32) << 20; out = (unsigned char *)malloc(out_cap);     /* (OUT_MB: the chunk builders cycle through 20 x 32 MiB) */
    for (rep = 0; rep < 3; rep++) {
        parser ps;
        double t0 = now(), dt;
        nrec = nbases = 0; out_len = 0;
        parser_init(&ps, on_rec, NULL);
        for (at = 0; at < (size_t)sb.st_size; at += blk) parser_feed(&ps, m + at, (size_t)sb.st_size - at < blk ? (size_t)sb.st_size - at : blk);
        parser_eof(&ps);
        dt = now() - t0;
        printf("%llu records, %llu bases of records >= 31: %.3f s = %.2f GB/s of file, %.2f Gbase/s\n", nrec, nbases, dt, sb.st_size / dt / 1e9, nbases / dt / 1e9);
        parser_free(&ps);
    }
    return 0;
}
