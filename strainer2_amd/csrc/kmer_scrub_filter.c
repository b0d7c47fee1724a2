/* kmer_scrub_filter: drop-in for the reference's scripts/kmer_scrub_filter.py (test/example.sh step 2);
 * everything is in libstrainer_kmer (skh_scrub_filter_main). */
#include <stdio.h>
#include "../../include/strainer_kmer.h"

int main(int argc, char **argv)
{
    return skh_scrub_filter_main(argc, argv, stdout, stderr);
}
