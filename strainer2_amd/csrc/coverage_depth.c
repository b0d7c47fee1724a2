/* coverage_depth: drop-in for the reference's scripts/coverage_depth.py (test/example.sh step 4);
 * everything is in libstrainer_kmer (skh_coverage_depth_main). */
#include <stdio.h>
#include "../../include/strainer_kmer.h"

int main(int argc, char **argv)
{
    return skh_coverage_depth_main(argc, argv, stdout, stderr);
}
