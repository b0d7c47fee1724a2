#!/usr/bin/env python3
"""Worst case for the counter atomics: every read is the SAME 150-base piece of the strain (4 M copies), so
120 counters take all the increments.  Device-resident pass time, and the counts checked (= copies)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import strainer2_amd as sk  # noqa: E402
from strainer2_amd import synth  # noqa: E402

N = int(os.environ.get("READS", "4000000"))
contigs = synth.make_strain()
sstream = synth.strain_stream(contigs)
ks = sk.Keyset.from_stream(sstream)
first = bytes(sstream[1000:1150])
assert b"N" not in first and b"\n" not in first
reads = np.frombuffer((first + b"\n") * N, dtype=np.uint8)
with sk.KmerContext(0) as ctx:
    ctx.load_keyset(ks, 4)
    dev = ctx.dev_alloc(reads.size)
    ctx.dev_upload(dev, reads)
    ctx.scan_device(dev, int(reads.size), 2)
    ctx.sync()
    c = ctx.counts(2)
    assert int(c.sum()) == 120 * N and int(c.max()) % N == 0, (int(c.sum()), int(c.max()))
    ctx.scan_timing(reset=True)
    for _ in range(3):
        ctx.scan_device(dev, int(reads.size), 2)
    ctx.sync()
    ms, n = ctx.scan_timing(reset=True)
print(f"{N} identical strain reads ({N * 150 / 1e9:.2f} Gbase): {ms / n:.2f} ms per pass = {N * 150 / (ms / n) / 1e6:.1f} Gbase/s; "
      f"all increments land on {int((c > 0).sum())} counters")
