#!/usr/bin/env python3
"""ONE big .gz through kmer_scrub_count's list scan (skh_scan_list on a resident table): the file's inflate runs on the
thread budget (sk_gzpar.h), its text is cut at checked record boundaries and parsed by helper threads (parse_gz_split) --
against one parser thread (SK_NO_SPLIT=1).  READS x 150 bp as FASTA.gz and FASTQ.gz (pigz-like level 4 via zlib)."""
import os
import subprocess
import sys
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
READS = int(os.environ.get("READS", "20000000"))
root = "/dev/shm/sk_onegz"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import strainer2_amd as sk
    from strainer2_amd import synth
    ks = sk.Keyset.from_stream(synth.strain_stream(synth.make_strain()))
    ctx = sk.KmerContext(0)
    ctx.load_keyset(ks, 4)
    for lst in sys.argv[2:]:
        best, nb = 1e9, 0
        for _ in range(3):
            ctx.zero_counts(1)
            t0 = time.perf_counter()
            nb = ctx.scan_list(os.path.join(root, lst), 1)
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        total = int(ctx.counts(1).astype(np.uint64).sum())
        print(f"  {os.environ.get('TAG', ''):28s} {lst:10s} {best:6.3f} s  {nb / best / 1e9:6.2f} Gbase/s   hits {total}", flush=True)
    sys.exit(0)

from strainer2_amd import synth  # noqa: E402
os.makedirs(root, exist_ok=True)
contigs = synth.make_strain()
t0 = time.time()
BLOCK = 2_000_000
for kind in ("fa", "fq"):
    co = zlib.compressobj(4, zlib.DEFLATED, 31)
    with open(os.path.join(root, f"one.{kind}.gz"), "wb") as f:
        for a0 in range(0, READS, BLOCK):
            m = min(BLOCK, READS - a0)
            reads, _ = synth.make_reads(contigs, m, 150, hit_frac=0.02, seed=synth.SEED + 7 + a0)
            rows = reads.reshape(m, 151)
            if kind == "fa":
                out = np.empty((m, 3 + 151), dtype=np.uint8)
                out[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
                out[:, 3:] = rows
            else:
                out = np.empty((m, 3 + 151 + 2 + 151), dtype=np.uint8)
                out[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
                out[:, 3:154] = rows
                out[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
                out[:, 156:306] = np.random.default_rng(a0).choice(np.frombuffer(b"FFFFFFFFFF::,#", dtype=np.uint8), size=(m, 150))
                out[:, 306] = 10
            f.write(co.compress(out.tobytes()))
        f.write(co.flush())
    open(os.path.join(root, f"{kind}.txt"), "w").write(os.path.join(root, f"one.{kind}.gz") + "\n")
print(f"{READS} reads x 150 bp = {READS * 150 / 1e9:.2f} Gbase; fa.gz {os.path.getsize(os.path.join(root, 'one.fa.gz')) / 1e9:.2f} GB, "
      f"fq.gz {os.path.getsize(os.path.join(root, 'one.fq.gz')) / 1e9:.2f} GB (written in {time.time() - t0:.0f} s)", flush=True)
for tag, env in (("one parser (the default)", {}), ("4 parser threads", {"SK_PARSE_THREADS": "4"}), ("8 parser threads", {"SK_PARSE_THREADS": "8"})):
    subprocess.run([sys.executable, __file__, "child", "fa.txt", "fq.txt"], env=dict(os.environ, TAG=tag, **env), check=True)
import shutil
shutil.rmtree(root, ignore_errors=True)
