/* strain_detect -- drop-in replacement for the reference program of the same name
 * (src/strain_detect.c): same flags, same messages, same gz result file (compare decompressed).
 * The work happens in libstrainer_kmer.so: host layer in C, every k-mer lookup in HIP on gfx950. */
#include <stdio.h>
#include "../../include/strainer_kmer.h"

int main(int argc, char **argv)
{
    return skh_strain_detect_main(argc, argv, stdout, stderr);
}
