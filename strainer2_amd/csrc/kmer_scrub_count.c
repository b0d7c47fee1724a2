/* kmer_scrub_count -- drop-in replacement for the reference program of the same name
 * (src/kmer_scrub_count.c:29-131): same flags (-r -A -B -C -p, and -d -h -u -H accepted), same
 * stdout TSV in the same row order, same progress file, same stderr texts and exit status.
 * All of the work happens in libstrainer_kmer.so (host layer in C, scan in HIP on gfx950). */
#include <stdio.h>
#include "../../include/strainer_kmer.h"

int main(int argc, char **argv)
{
    static char obuf[1 << 20];
    setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    return skh_kmer_scrub_count_main(argc, argv, stdout, stderr);
}
