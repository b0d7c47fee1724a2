#!/usr/bin/env python3
"""One record of 50 Mbase on a single line (the strain ten times over, as a gzipped one-line FASTA) through
kmer_scrub_count's -A list: every row's pangenome_count must be exactly 10 x its reference_count.  Exercises the record
cutting with k-1 overlap at full scale, the parser on a 50 MB line, and the several-thread inflate on FASTA."""
import gzip
import os
import subprocess
import sys
import time

import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402

w = "/tmp/sk_long"
os.makedirs(w, exist_ok=True)
contigs = synth.make_strain(n_enn=0, contig_bp=5_000_000)           # one contig of 5 Mbp, no N
open(f"{w}/strain.fa", "wb").write(synth.strain_fasta(contigs))
seq = contigs[0].tobytes()
with gzip.open(f"{w}/chr.fa.gz", "wb", compresslevel=1) as f:
    f.write(b">chr\n" + seq * 10 + b"\n")
open(f"{w}/A.txt", "w").write(f"{w}/chr.fa.gz\n")
open(f"{w}/few.fa", "wb").write(b">r\n" + seq[:200] + b"\n")
open(f"{w}/B.txt", "w").write(f"{w}/few.fa\n")                  # (-B is not optional)
t = time.time()
with open(f"{w}/out.tsv", "wb") as f:
    subprocess.run([os.path.join(REPO, "strainer2_amd", "bin", "kmer_scrub_count"), "-r", f"{w}/strain.fa", "-A", f"{w}/A.txt", "-B", f"{w}/B.txt"], stdout=f, check=True)
dt = time.time() - t
d = pd.read_csv(f"{w}/out.tsv", sep="\t")
ok = bool((d["pangenome_count"] == 10 * d["reference_count"]).all())
print(f"{len(d)} rows, {os.path.getsize(w + '/chr.fa.gz') / 1e6:.0f} MB gz, {dt:.2f} s; pangenome_count == 10 x reference_count in every row: {ok}")
assert ok
