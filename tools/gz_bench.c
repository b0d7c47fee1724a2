/* gz_bench.c -- inflate rate of strainer2_amd/csrc/sk_gzfast.h (1 thread) and sk_gzpar.h (several threads on one
 * member) on a .gz file, host only, bytes thrown away:
 *     gcc -O2 -o /tmp/gz_bench tools/gz_bench.c -lpthread && /tmp/gz_bench reads.fq.gz 1 2 4 8 16
 * (numbers in DESIGN.md; the first run of a thread count pays its page faults, so every count is run twice) */
#define _GNU_SOURCE
#include <stdio.h>
#include <time.h>
#include "../strainer2_amd/csrc/sk_gzpar.h"

static size_t got;
static unsigned acc;
static int sink(void *u, const unsigned char *d, size_t n) { (void)u; got += n; acc += d[0] + d[n - 1]; return 0; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    struct stat st;
    unsigned char *m;
    int fd, i, rep;
    if (argc < 3 || (fd = open(argv[1], O_RDONLY)) < 0 || fstat(fd, &st)) { fprintf(stderr, "usage: gz_bench file.gz threads...\n"); return 2; }
    m = (unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    if (m == MAP_FAILED) return 2;
    for (i = 2; i < argc; i++)
        for (rep = 0; rep < 2; rep++) {
            double t0, dt;
            int rc;
            got = 0;
            skzq_stat_direct = skzq_stat_gap = skzq_stat_again = 0;
            t0 = now();
            rc = skzq_decode_memory(m, (size_t)st.st_size, atoi(argv[i]), 0, sink, NULL);
            dt = now() - t0;
            printf("threads %2d: rc %d, %zu -> %zu bytes in %.3f s = %.0f MB/s of text  (segments: %llu as guessed, %llu after a gap, %llu decoded again)\n",
                   atoi(argv[i]), rc, (size_t)st.st_size, got, dt, (double)got / dt / 1e6,
                   (unsigned long long)skzq_stat_direct, (unsigned long long)skzq_stat_gap, (unsigned long long)skzq_stat_again);
        }
    return (int)(acc & 0);
}
