#!/usr/bin/env python3
"""strain_detect with many strains resident (cfg 5 shape, scaled): NSTRAINS synthetic strains of STRAIN_BP
each, 1 % of their k-mers informative, one SE FASTA of READS x 150 bp reads with 2 % of the reads drawn
from the strains.  Wall clock of ONE `strain_detect -S list` pass against NSTRAINS separate runs of the
same program (what the reference workflow does), outputs compared.  Prints one JSON line."""
import gzip
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402
import strainer2_amd as sk  # noqa: E402

NSTRAINS = int(os.environ.get("NSTRAINS", "8"))
STRAIN_BP = int(os.environ.get("STRAIN_BP", "5000000"))
READS = int(os.environ.get("READS", "2000000"))
SEPARATE = int(os.environ.get("SEPARATE", str(NSTRAINS)))       # how many of the separate runs to time
work = os.environ.get("WORK", "/tmp/sk_sdm")
os.makedirs(work, exist_ok=True)
rng = np.random.default_rng(11)
t0 = time.time()
strains = []
with open(os.path.join(work, "strains.txt"), "w") as lst:
    for s in range(NSTRAINS):
        g = synth._rand_bases(rng, STRAIN_BP)
        strains.append(g)
        with open(os.path.join(work, f"s{s}.fa"), "wb") as f:
            f.write(b">s%d\n" % s + g.tobytes() + b"\n")
        ks = sk.Keyset.from_stream(g.tobytes() + b"\n")
        pick = np.sort(rng.choice(ks.nrows, size=ks.nrows // 100, replace=False))
        with open(os.path.join(work, f"s{s}.inf"), "wb") as f:
            f.write(b"\n".join(ks.key(int(i)) for i in pick) + b"\n")
        ks.close()
        lst.write(f"{work}/s{s}.fa\t{work}/s{s}.inf\t{work}/multi{s}.gz\n")
# the metagenome, written a block of reads at a time (10 Gbase = 66.7 M reads = 10.3 GB of FASTA)
BLOCK = 2_000_000
head = np.frombuffer(b">r\n", dtype=np.uint8)
with open(os.path.join(work, "reads.fa"), "wb") as f:
    for a0 in range(0, READS, BLOCK):
        m = min(BLOCK, READS - a0)
        blk = synth._rand_bases(rng, m * 150).reshape(m, 150)
        for i in np.flatnonzero(rng.random(m) < 0.02):
            g = strains[int(rng.integers(0, NSTRAINS))]
            p0 = int(rng.integers(0, STRAIN_BP - 150))
            blk[i] = g[p0:p0 + 150]
        fa = np.empty((m, 3 + 151), dtype=np.uint8)
        fa[:, :3] = head
        fa[:, 3:153] = blk
        fa[:, 153] = 10
        fa.tofile(f)
print(f"inputs ready in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)

exe = os.path.join(REPO, "strainer2_amd", "bin", "strain_detect")
t = time.time()
subprocess.run([exe, "-S", os.path.join(work, "strains.txt"), "-b", os.path.join(work, "reads.fa"), "-t", "SE"], check=True)
t_multi = time.time() - t
t_members = None
if os.environ.get("ALSO_NO_UNION"):                  # the same pass strain by strain (one launch per strain and batch)
    import hashlib
    md5 = [hashlib.md5(gzip.open(f"{work}/multi{s}.gz").read()).hexdigest() for s in range(NSTRAINS)]
    t = time.time()
    subprocess.run([exe, "-S", os.path.join(work, "strains.txt"), "-b", os.path.join(work, "reads.fa"), "-t", "SE"], check=True,
                   env=dict(os.environ, SK_SD_NO_UNION="1"))
    t_members = time.time() - t
    assert md5 == [hashlib.md5(gzip.open(f"{work}/multi{s}.gz").read()).hexdigest() for s in range(NSTRAINS)], "union and member-by-member outputs differ"
t_sep = []
same = True
for s in range(SEPARATE):
    t = time.time()
    subprocess.run([exe, "-r", f"{work}/s{s}.fa", "-a", f"{work}/s{s}.inf", "-b", os.path.join(work, "reads.fa"), "-t", "SE",
                    "-o", f"{work}/single{s}.gz"], check=True)
    t_sep.append(time.time() - t)
    same = same and gzip.open(f"{work}/single{s}.gz").read() == gzip.open(f"{work}/multi{s}.gz").read()
lines = sum(1 for s in range(NSTRAINS) for _ in gzip.open(f"{work}/multi{s}.gz"))
bases = READS * 150
print(json.dumps({
    "workload": f"{NSTRAINS} strains x {STRAIN_BP} bp, 1 % informative; {READS} x 150 bp SE reads ({bases / 1e9:.2f} Gbase), 2 % from the strains",
    "one_pass_all_strains_s": round(t_multi, 2),
    "one_pass_strain_by_strain_s": round(t_members, 2) if t_members else None,
    "separate_runs_timed": SEPARATE,
    "separate_run_mean_s": round(float(np.mean(t_sep)), 2) if t_sep else None,
    "separate_runs_total_s_extrapolated": round(float(np.mean(t_sep)) * NSTRAINS, 2) if t_sep else None,
    "strain_x_bases_per_s_one_pass": round(NSTRAINS * bases / t_multi),
    "output_lines": lines, "outputs_identical_to_separate_runs": same}))
