/* strain_detect -- drop-in replacement for the reference program of the same name
 * (src/strain_detect.c): same flags, same messages, same gz result file (compare decompressed).
 * The work happens in libstrainer_kmer.so: host layer in C, every k-mer lookup in HIP on gfx950. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/strainer_kmer.h"

int main(int argc, char **argv)
{
    int rc;
    /* this process ends with the run: the library may leave its device contexts and key sets to the end of the process
     * instead of taking them apart (SK_LEAK_AT_EXIT=0: the orderly way, for leak checkers) */
    setenv("SK_LEAK_AT_EXIT", "1", 0);
    rc = skh_strain_detect_main(argc, argv, stdout, stderr);
    if (strcmp(getenv("SK_LEAK_AT_EXIT"), "0")) { fflush(stdout); fflush(stderr); _exit(rc); }
    return rc;
}
