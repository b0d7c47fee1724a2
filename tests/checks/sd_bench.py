#!/usr/bin/env python3
"""strain_detect end to end on synthetic data (5 Mbp strain, 1 % informative k-mers, SE FASTA of READS
reads with 2 % strain reads): wall clock of strainer2_amd/bin/strain_detect.  For DESIGN.md."""
import gzip
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402
import strainer2_amd as sk  # noqa: E402

READS = int(os.environ.get("READS", "4000000"))
work = os.environ.get("WORK", "/tmp/sk_sd")
os.makedirs(work, exist_ok=True)
contigs = synth.make_strain(total_bp=int(os.environ.get("STRAIN_BP", "5000000")))
open(os.path.join(work, "strain.fa"), "wb").write(synth.strain_fasta(contigs))
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
keys = ks.keys()
rng = np.random.default_rng(7)
pick = rng.choice(len(keys), size=len(keys) // 100, replace=False)
open(os.path.join(work, "inf.txt"), "wb").write(b"#informative\n" + b"\n".join(keys[i] for i in sorted(pick)) + b"\n")
stream, _ = synth.make_reads(contigs, READS, seed=synth.SEED + 500, hit_frac=float(os.environ.get("HIT_FRAC", "0.02")))
rows = stream.reshape(READS, 151)[:, :150]
PE = bool(os.environ.get("PE"))          # PE=1: the reads as a pair of files (mates = even / odd reads), -t PE
GZ = bool(os.environ.get("GZ"))          # GZ=1: the reads as one .gz file (single-file inflate is then what is timed)
reads_name = "reads.fa.gz" if GZ else "reads.fa"
def write_reads(name, sel):
    with (gzip.open(os.path.join(work, name), "wb", compresslevel=4) if GZ else open(os.path.join(work, name), "wb")) as f:
        f.write(b"".join(b">r%d\n%s\n" % (j, rows[j].tobytes()) for j in sel))


if PE:
    mate2 = reads_name.replace("reads", "reads_2")
    write_reads(reads_name, range(0, READS, 2))
    write_reads(mate2, range(1, READS, 2))
    files = ["-b", os.path.join(work, reads_name), "-c", os.path.join(work, mate2), "-t", "PE"]
else:
    write_reads(reads_name, range(READS))
    files = ["-b", os.path.join(work, reads_name), "-t", "SE"]
exe = os.path.join(REPO, "strainer2_amd", "bin", "strain_detect")
for gzt in (os.environ.get("GZ_THREADS", "").split(",") if GZ else [""]):
    env = dict(os.environ, SK_GZ_THREADS=gzt) if gzt else dict(os.environ)
    t = time.time()
    subprocess.run([exe, "-r", os.path.join(work, "strain.fa"), "-a", os.path.join(work, "inf.txt")] + files +
                   ["-o", os.path.join(work, "hits.gz")], check=True, env=env)
    dt = time.time() - t
    n = sum(1 for _ in gzip.open(os.path.join(work, "hits.gz")))
    print(f"strain_detect{' PE' if PE else ''}{' SK_GZ_THREADS=' + gzt if gzt else ''}: {READS} reads ({READS * 150 / 1e9:.2f} Gbase, {reads_name}) in {dt:.2f} s wall => "
          f"{READS * 150 / dt / 1e9:.3f} Gbase/s end to end; {n} output lines", flush=True)
ref = os.path.join(REPO, "oracle", "_ref", "strain_detect")
if os.path.exists(ref) and os.environ.get("WITH_REF"):
    t = time.time()
    subprocess.run([ref, "-r", os.path.join(work, "strain.fa"), "-a", os.path.join(work, "inf.txt")] + files +
                   ["-o", os.path.join(work, "ref_hits.gz")], check=True)
    dr = time.time() - t
    same = gzip.open(os.path.join(work, "hits.gz")).read() == gzip.open(os.path.join(work, "ref_hits.gz")).read()
    print(f"reference strain_detect: {dr:.2f} s wall; outputs identical: {same}")
