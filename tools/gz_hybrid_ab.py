#!/usr/bin/env python3
"""tools/gz_hybrid_ab.py -- EXPERIMENT (make EXPERIMENTS=1 build under build_exp/): a -B list of NFILES .gz FASTQ files through kmer_scrub_count
with the host's decode threads alone, and with k extra threads that feed the device-side gzip decoder beside them (SK_GPU_INFLATE=1,
SK_GPU_INFLATE_WORKERS=k).  Prints wall, the program's timing lines and whether the table is the same.  GPU box.
  NFILES=32 READS=1000000 python3 tools/gz_hybrid_ab.py"""
import gzip
import hashlib
import multiprocessing as mp
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402

NFILES = int(os.environ.get("NFILES", "32"))
READS = int(os.environ.get("READS", "1000000"))
WORK = os.environ.get("WORK", "/dev/shm/sk_gz_hybrid")


def write_one(i):
    contigs = synth.make_strain(total_bp=5_000_000)
    stream, _ = synth.make_reads(contigs, READS, seed=synth.SEED + 100 + i)
    rows = stream.reshape(READS, 151)[:, :150]
    q = np.random.default_rng(1000 + i).choice(np.frombuffer(b"FFFFFFFFFF::,#", dtype=np.uint8), size=(READS, 150))
    rec = np.empty((READS, 12 + 151 + 2 + 151), dtype=np.uint8)
    rec[:, :12] = np.frombuffer(b"@r0000000000", dtype=np.uint8)
    rec[:, 11] = 10
    rec[:, 12:162] = rows
    rec[:, 162] = 10
    rec[:, 163] = ord("+")
    rec[:, 164] = 10
    rec[:, 165:315] = q
    rec[:, 315] = 10
    p = os.path.join(WORK, f"reads{i}.fq.gz")
    with gzip.open(p, "wb", compresslevel=4) as f:
        f.write(rec.tobytes())
    return p


def main():
    os.makedirs(WORK, exist_ok=True)
    t0 = time.time()
    with mp.Pool(16) as pool:
        names = pool.map(write_one, range(NFILES))
    contigs = synth.make_strain(total_bp=5_000_000)
    open(os.path.join(WORK, "strain.fa"), "wb").write(synth.strain_fasta(contigs))
    open(os.path.join(WORK, "B.txt"), "w").write("\n".join(names) + "\n")
    open(os.path.join(WORK, "A.txt"), "w").write(os.path.join(WORK, "strain.fa") + "\n")
    gz_bytes = sum(os.path.getsize(n) for n in names)
    print(f"{NFILES} files x {READS} reads, {gz_bytes / 1e6:.0f} MB of .gz, written in {time.time() - t0:.0f} s", flush=True)
    exe = os.path.join(REPO, "build_exp", "bin", "kmer_scrub_count")
    argv = [exe, "-r", os.path.join(WORK, "strain.fa"), "-A", os.path.join(WORK, "A.txt"), "-B", os.path.join(WORK, "B.txt")]
    md5s = {}
    for name, env in [("host only", {}), ("host only", {})] + [(f"+{k} device feeders", {"SK_GPU_INFLATE": "1", "SK_GPU_INFLATE_WORKERS": str(k)}) for k in (2, 4, 8)] + [("host only", {})]:
        t1 = time.time()
        p = subprocess.run(argv, capture_output=True, env=dict(os.environ, SK_TIMING="1", **env))
        wall = time.time() - t1
        md5 = hashlib.md5(p.stdout).hexdigest()
        md5s.setdefault(md5, []).append(name)
        timing = [ln for ln in p.stderr.decode().split("\n") if ln.startswith("kmer_scrub_count timing")]
        print(f"{name:22s} rc {p.returncode} wall {wall:.2f} s  {NFILES * READS * 150 / wall / 1e9:.2f} Gbase/s  {timing[-1] if timing else ''}", flush=True)
    print("tables identical:", len(md5s) == 1)
    import shutil
    shutil.rmtree(WORK, ignore_errors=True)


if __name__ == "__main__":
    main()
