#!/usr/bin/env python3
"""tools/sd_many_files.py -- strain_detect on the SAME reads as ONE file and as a -B list of NFILES files: what a file costs.  GPU box.
  NFILES=64 READS=2000000 STRAINS=4 python3 tools/sd_many_files.py"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import cfg5  # noqa: E402

NFILES = int(os.environ.get("NFILES", "64"))
READS = int(os.environ.get("READS", "2000000"))
WORK = os.environ.get("WORK", "/dev/shm/sk_sd_files")


def main():
    paths = cfg5.write_all(WORK, procs=16, prefix_reads=READS, only_prefix=True)
    pre = open(os.path.join(WORK, "prefix.fa"), "rb").read()
    recs = pre.split(b">")[1:]
    per = (len(recs) + NFILES - 1) // NFILES
    names = []
    for i in range(NFILES):
        p = os.path.join(WORK, f"part{i:03d}.fa")
        open(p, "wb").write(b"".join(b">" + r for r in recs[i * per:(i + 1) * per]))
        names.append(p)
    open(os.path.join(WORK, "B_many.txt"), "w").write("".join(f"SE\t{n}\n" for n in names))
    open(os.path.join(WORK, "B_one.txt"), "w").write(f"SE\t{os.path.join(WORK, 'prefix.fa')}\n")
    exe = os.path.join(REPO, "strainer2_amd", "bin", "strain_detect")
    for name, lst in (("one file", "B_one.txt"), (f"{NFILES} files", "B_many.txt"), ("one file", "B_one.txt"), (f"{NFILES} files", "B_many.txt")):
        t0 = time.time()
        p = subprocess.run([exe, "-S", paths["strains_prefix"] if "strains_prefix" in paths else os.path.join(WORK, "strains_prefix.txt"), "-B", os.path.join(WORK, lst)],
                           cwd=WORK, capture_output=True, env=dict(os.environ, SK_SD_TIMING="1"))
        wall = time.time() - t0
        t = [ln for ln in p.stderr.decode().split("\n") if "setup" in ln]
        print(f"{name:10s} rc {p.returncode} wall {wall:.2f} s  {t[0][21:] if t else p.stderr.decode()[-300:]}", flush=True)
    import shutil
    shutil.rmtree(WORK, ignore_errors=True)


if __name__ == "__main__":
    main()
