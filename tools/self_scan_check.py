#!/usr/bin/env python3
"""Size-independent exactness check through the program: a strain scanned against a copy of its own FASTA must show
pangenome_count == metagenome_count == reference_count in every row.  Shapes that no BASELINE config has:
fragmented assemblies (contigs of 31..100 bases), N-riddled and IUPAC-riddled strains, lower case.
usage: self_scan_check.py  (runs a fixed list of shapes; STRAIN_BP scales them)"""
import os
import subprocess
import sys
import time

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(REPO, "strainer2_amd", "bin", "kmer_scrub_count")
bp = int(os.environ.get("STRAIN_BP", "3000000"))
w = "/tmp/sk_self"
os.makedirs(w, exist_ok=True)
rng = np.random.default_rng(99)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0
shapes = [("IUPAC letters every ~%d bases" % round(1 / float(os.environ["IUPAC_RATE"])), 50_000, 0, float(os.environ["IUPAC_RATE"]), 0)] if os.environ.get("IUPAC_RATE") else None
for name, contig, enn, iupac, lower in shapes or (("contigs of 100 kb", 100_000, 0, 0, 0), ("contigs of 100", 100, 0, 0, 0), ("contigs of 31", 31, 0, 0, 0),
                                        ("contigs of 33..45", -1, 0, 0, 0), ("an N every ~150 bases", 50_000, 1 / 150, 0, 0),
                                        ("IUPAC letters every ~500 bases", 50_000, 0, 1 / 500, 0), ("lower case, N and IUPAC", 20_000, 1 / 2000, 1 / 2000, 1)):
    seq = acgt[rng.integers(0, 4, bp)]
    if enn:
        seq[rng.random(bp) < enn] = ord("N")
    if iupac:
        m = rng.random(bp) < iupac
        seq[m] = np.frombuffer(b"RYKMSWBDHVU", dtype=np.uint8)[rng.integers(0, 11, int(m.sum()))]
    if lower:
        m = rng.random(bp) < 0.3
        seq[m] |= 0x20
    out, pos, i = [], 0, 0
    while pos < bp:
        n = contig if contig > 0 else int(rng.integers(33, 46))
        out.append(b">c%d\n%s\n" % (i, seq[pos:pos + n].tobytes()))
        pos += n
        i += 1
    fa = b"".join(out)
    open(f"{w}/strain.fa", "wb").write(fa)
    open(f"{w}/copy.fa", "wb").write(fa)
    open(f"{w}/L.txt", "w").write(f"{w}/copy.fa\n")
    t = time.time()
    p = subprocess.run([exe, "-r", f"{w}/strain.fa", "-A", f"{w}/L.txt", "-B", f"{w}/L.txt"], capture_output=True)
    dt = time.time() - t
    if p.returncode:
        print(f"{name}: exit {p.returncode}: {p.stderr.decode()[-300:]}")
        bad += 1
        continue
    open(f"{w}/out.tsv", "wb").write(p.stdout)
    d = pd.read_csv(f"{w}/out.tsv", sep="\t")
    ok = bool(((d["pangenome_count"] == d["reference_count"]) & (d["metagenome_count"] == d["reference_count"])).all())
    print(f"{name}: {i} contigs, {len(d)} rows ({int((d['reference_count'] > 1).sum())} of them more than once in the strain), {dt:.2f} s, "
          f"every row counted as often as it occurs: {ok}", flush=True)
    if os.environ.get("SK_TIMING"):
        print("   " + " | ".join(line for line in p.stderr.decode().splitlines() if "timing" in line or "key set" in line))
    bad += not ok
sys.exit(bad)
