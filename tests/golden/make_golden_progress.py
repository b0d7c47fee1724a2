#!/usr/bin/env python3
"""Two more golden cases for the list walk's failure path (VERDICT r02 weak #8a), made like the others by
running the UNMODIFIED reference binary (oracle/_ref/kmer_scrub_count) in the build container:

  progress_missing     -p with a file that cannot be read in the MIDDLE of the -B list: the reference writes a
                       list line to the progress file just before it opens that file (src/genome_compare.c:167-172)
                       and exits in the open (:195-198) -- the progress file ends with the failing line
  skip_after_missing   a -C list whose unreadable line comes BEFORE the line that equals -r: the reference never
                       reaches the skip message (:138-141)

Kept apart from make_golden.py so that its random stream (and with it every older fixture) stays as it is.

    make -C oracle ref && python tests/golden/make_golden_progress.py
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    if not os.path.exists(mg.REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    rng = random.Random(0x9A07)
    s = mg.rand_dna(rng, 300)
    reads = [s[a:a + 60] for a in (0, 40, 100, 170, 230)]
    for name in ("progress_missing", "skip_after_missing"):
        d = os.path.join(mg.CASES, name)
        mg.write(os.path.join(d, "strain.fa"), ">s\n" + mg.wrap(s, 70) + "\n")
        mg.write(os.path.join(d, "g1.fa"), ">g\n" + s[20:200] + "\n")
        for i in range(4):
            mg.write(os.path.join(d, f"m{i}.fa"), "".join(f">r{j}\n{r}\n" for j, r in enumerate(reads[i:i + 2])))
        mg.write(os.path.join(d, "A.txt"), "g1.fa\n")
    d = os.path.join(mg.CASES, "progress_missing")
    mg.write(os.path.join(d, "B.txt"), "m0.fa\nm1.fa\nnope.fa\nm2.fa\nm3.fa\n")
    mg.run_case("progress_missing", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-p", "progress.txt"],
                "the progress file ends with the line of the file that could not be read; no table, exit 1")
    d = os.path.join(mg.CASES, "skip_after_missing")
    mg.write(os.path.join(d, "B.txt"), "m0.fa\n")
    mg.write(os.path.join(d, "C.txt"), "strain.fa\nm1.fa\nnope.fa\nstrain.fa\nm2.fa\n")
    mg.run_case("skip_after_missing", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-C", "C.txt", "-p", "progress.txt"],
                "one skip message before the unreadable line, none for the -r line after it; progress ends with the failing line")


if __name__ == "__main__":
    main()
