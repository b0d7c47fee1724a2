#!/usr/bin/env python3
"""A strain bigger than anything in the test suite (STRAIN_BP, default 20 Mbp): key set, table load, one scan of
READS reads against the oracle-free invariants (hits of strain-drawn reads), each stage timed and printed at once."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import strainer2_amd as sk  # noqa: E402
from strainer2_amd import synth  # noqa: E402

bp = int(os.environ.get("STRAIN_BP", "20000000"))
reads = int(os.environ.get("READS", "1000000"))
t = time.time()
contigs = synth.make_strain(total_bp=bp)
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
print(f"key set of {bp} bp: {ks.nrows} rows in {time.time() - t:.2f} s", flush=True)
t = time.time()
c = sk.KmerContext(0)
c.load_keyset(ks, 4)
print(f"table load: {time.time() - t:.2f} s", flush=True)
stream, nbases = synth.make_reads(contigs, reads, seed=synth.SEED + 1, hit_frac=0.02)
t = time.time()
c.scan_stream(stream.tobytes() if hasattr(stream, "tobytes") else stream, 2)
counts = c.counts(2)
print(f"scan of {reads} reads: {time.time() - t:.2f} s, {int(counts.sum())} hits in {nbases} bases", flush=True)
c.close()
