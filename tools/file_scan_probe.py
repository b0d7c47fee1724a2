#!/usr/bin/env python3
"""Where does a file-fed list scan spend its time?  One plain FASTQ / FASTA file (READS x 150 bp) under /dev/shm, scanned
through skh_scan_list with SK_THREADS = 1, 2, 4, 8, 16 (each in a fresh process: the thread budget is read once), and the
same bytes as 16 files.  Prints one line per case."""
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

READS = int(os.environ.get("READS", "4000000"))
root = "/dev/shm/sk_probe"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import strainer2_amd as sk
    from strainer2_amd import synth
    ks = sk.Keyset.from_stream(synth.strain_stream(synth.make_strain()))
    ctx = sk.KmerContext(0)
    ctx.load_keyset(ks, 4)
    for lst in sys.argv[2:]:
        t0 = time.perf_counter()
        ctx.scan_list(os.path.join(root, lst), 1)         # first call: page cache, page-locked buffers, threads
        ctx.sync()
        cold = time.perf_counter() - t0
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            nb = ctx.scan_list(os.path.join(root, lst), 1)
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        size = sum(os.path.getsize(l.strip()) for l in open(os.path.join(root, lst)))
        print(f"  SK_THREADS={os.environ.get('SK_THREADS', '-'):>2} {lst:12s} {best * 1e3:8.1f} ms  {nb / best / 1e9:6.2f} Gbase/s  {size / best / 1e9:6.2f} GB/s of text   (first call {cold * 1e3:7.1f} ms)", flush=True)
    sys.exit(0)

from strainer2_amd import synth  # noqa: E402
os.makedirs(root, exist_ok=True)
reads, nbases = synth.make_reads(synth.make_strain(), READS, 150, hit_frac=0.02, seed=synth.SEED + 1)
rows = reads.reshape(READS, 151)
fq = np.empty((READS, 3 + 151 + 2 + 151), dtype=np.uint8)
fq[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
fq[:, 3:154] = rows
fq[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
fq[:, 156:306] = ord("I")
fq[:, 306] = 10
fq.tofile(os.path.join(root, "one.fq"))
per = READS // 16
for k in range(16):
    fq[k * per:(k + 1) * per].tofile(os.path.join(root, f"p{k}.fq"))
fa = np.empty((READS, 3 + 151), dtype=np.uint8)
fa[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
fa[:, 3:] = rows
fa.tofile(os.path.join(root, "one.fa"))
open(os.path.join(root, "one_fq.txt"), "w").write(os.path.join(root, "one.fq") + "\n")
open(os.path.join(root, "one_fa.txt"), "w").write(os.path.join(root, "one.fa") + "\n")
open(os.path.join(root, "many_fq.txt"), "w").write("".join(os.path.join(root, f"p{k}.fq") + "\n" for k in range(16)))
print(f"{READS} reads x 150 bp = {nbases / 1e9:.2f} Gbase; FASTQ {os.path.getsize(os.path.join(root, 'one.fq')) / 1e9:.2f} GB, FASTA {os.path.getsize(os.path.join(root, 'one.fa')) / 1e9:.2f} GB", flush=True)
for th in (1, 2, 4, 8, 16):
    subprocess.run([sys.executable, __file__, "child", "one_fq.txt", "one_fa.txt", "many_fq.txt"], env=dict(os.environ, SK_THREADS=str(th)), check=True)
for mib in (4, 8, 16):
    print(f" SK_CHUNK_BYTES={mib} MiB", flush=True)
    subprocess.run([sys.executable, __file__, "child", "one_fq.txt", "one_fa.txt", "many_fq.txt"], env=dict(os.environ, SK_THREADS="16", SK_CHUNK_BYTES=str(mib << 20)), check=True)
import shutil
shutil.rmtree(root, ignore_errors=True)
