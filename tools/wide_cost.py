#!/usr/bin/env python3
"""Cost of the byte-string ("wide") path: the cfg-2 batch with NO byte outside ACGTN, with ONE such byte, and
with one read in a thousand carrying an IUPAC letter.  Device-resident passes, kernel time from HIP events."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import strainer2_amd as sk  # noqa: E402
from strainer2_amd import synth  # noqa: E402

READS = int(os.environ.get("READS", "10000000"))
contigs = synth.make_strain()
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
reads, nb = synth.make_reads(contigs, READS, 150, seed=synth.SEED + 1)
with sk.KmerContext(0) as ctx:
    ctx.load_keyset(ks, 4)
    dev = ctx.dev_alloc(reads.size)
    for label, edit in (("no odd byte", None), ("one odd byte", 1), ("one read in 1000 with an IUPAC letter", READS // 1000)):
        r = reads.copy()
        if edit:
            rng = np.random.default_rng(3)
            idx = rng.choice(READS, size=edit, replace=False).astype(np.int64) * 151 + 75
            r[idx] = ord("R")
        ctx.dev_upload(dev, r)
        ctx.scan_device(dev, int(r.size), 2)
        ctx.sync()
        ctx.scan_timing(reset=True)
        for _ in range(5):
            ctx.scan_device(dev, int(r.size), 2)
        ctx.sync()
        ms, n = ctx.scan_timing(reset=True)
        print(f"{label}: {ms / n:.3f} ms per {nb / 1e9:.2f} Gbase pass (main kernel only; see rocprof for sk_scan_wide)", flush=True)
