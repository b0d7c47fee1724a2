#!/usr/bin/env python3
"""Differential check at a size between the fixtures and the benchmarks, on data where the byte-string path is BUSY: a
300 kb strain and 40,000 reads with an IUPAC letter / U / N / lower case every few hundred bases, both programs against
the oracle programs (oracle/kso_oracle, oracle/ksd_oracle; on the GPU box these are the prebuilt checkers).  Test tool."""
import gzip
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, O = os.path.join(REPO, "strainer2_amd", "bin"), os.path.join(REPO, "oracle")
w = "/tmp/sk_iupac"
os.makedirs(w, exist_ok=True)
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
odd = np.frombuffer(b"RYKMSWBDHVUNnacgtu", dtype=np.uint8)


def sprinkle(a, rate):
    m = rng.random(a.shape) < rate
    a[m] = odd[rng.integers(0, len(odd), int(m.sum()))]
    return a


n = 300_000
strain = sprinkle(acgt[rng.integers(0, 4, n)], 1 / 400)
open(f"{w}/strain.fa", "wb").write(b"".join(b">c%d\n%s\n" % (i, strain[i:i + 50_000].tobytes()) for i in range(0, n, 50_000)))
nr = 40_000
starts = rng.integers(0, n - 150, nr)
reads = strain[starts[:, None] + np.arange(150)[None, :]].copy()
reads[rng.random(nr) < 0.5] = acgt[rng.integers(0, 4, 150)]              # half of them not from the strain
reads = sprinkle(reads, 1 / 600)
# SHORT=1: a third of the reads cut to 5..60 bases -- reads shorter than k re-use (and re-emit) the tallies of the read before
lens = np.full(nr, 150)
if os.environ.get("SHORT"):
    cut = rng.random(nr) < 0.33
    lens[cut] = rng.integers(5, 61, int(cut.sum()))
for name, sel in (("g.fa", slice(0, 10_000)), ("m_1.fa", slice(10_000, 25_000)), ("m_2.fa", slice(25_000, 40_000))):
    open(f"{w}/{name}", "wb").write(b"".join(b">r%d\n%s\n" % (j, reads[j, :lens[j]].tobytes()) for j in range(sel.start, sel.stop)))
open(f"{w}/il.fa", "wb").write(b"".join(b">r%d\n%s\n>q%d\n%s\n" % (j, reads[10_000 + j, :lens[10_000 + j]].tobytes(), j, reads[25_000 + j, :lens[25_000 + j]].tobytes()) for j in range(15_000)))
open(f"{w}/A.txt", "w").write(f"{w}/g.fa\n")
open(f"{w}/B.txt", "w").write(f"{w}/m_1.fa\n{w}/m_2.fa\n")
bad = 0
t = time.time()
argv = ["-r", f"{w}/strain.fa", "-A", f"{w}/A.txt", "-B", f"{w}/B.txt"]
a = subprocess.run([os.path.join(B, "kmer_scrub_count")] + argv, capture_output=True)
b = subprocess.run([os.path.join(O, "kso_oracle")] + argv, capture_output=True)
same = (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr)
rows = a.stdout.splitlines()
wide = sum(1 for r in rows[1:] if any(c not in b"ACGT" for c in r.split(b"\t")[0]))
print(f"kmer_scrub_count vs oracle: {len(rows) - 1} rows ({wide} byte-string keys), hits {sum(int(r.split(bytes([9]))[2]) + int(r.split(bytes([9]))[3]) for r in rows[1:])}: {'identical' if same else 'DIFFERENT'}", flush=True)
bad += not same
# informative k-mers: every 7th row, byte-string keys included
inf = [r.split(b"\t")[0] for r in rows[1::7]]
open(f"{w}/inf.txt", "wb").write(b"#kmer\n" + b"\n".join(inf) + b"\n")
for mode, files in (("PE", ["-b", f"{w}/m_1.fa", "-c", f"{w}/m_2.fa"]), ("SE", ["-b", f"{w}/m_1.fa"]), ("PEI", ["-b", f"{w}/il.fa"])):
    outs = []
    for exe, tag in ((os.path.join(B, "strain_detect"), "gpu"), (os.path.join(O, "ksd_oracle"), "oracle")):
        o = f"{w}/s_m_{tag}{mode}.kmer_hits.gz"
        p = subprocess.run([exe, "-r", f"{w}/strain.fa", "-a", f"{w}/inf.txt"] + files + ["-t", mode, "-o", o], capture_output=True)
        outs.append((p.returncode, p.stdout, gzip.open(o, "rb").read() if os.path.exists(o) else b""))
    same = outs[0] == outs[1]
    print(f"strain_detect {mode} vs oracle: {len(outs[0][2].splitlines())} output lines: {'identical' if same else 'DIFFERENT'}", flush=True)
    bad += not same
print(f"{time.time() - t:.1f} s")
sys.exit(bad)
