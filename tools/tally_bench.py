#!/usr/bin/env python3
"""Kernel rate of the TALLY mode (strain_detect): sk_tally_batch over cfg-2-like reads
(5 Mbp strain, 1 % of its k-mers informative, 2 % strain reads) in 48 MiB batches; kernel time from the
library's HIP events.  Prints one JSON line (for DESIGN.md / profiles/, not the driver's bench)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import strainer2_amd as sk  # noqa: E402
from strainer2_amd import synth  # noqa: E402

READS = int(os.environ.get("READS", "4000000"))
contigs = synth.make_strain()
ks = sk.Keyset.from_stream(synth.strain_stream(contigs), default_val=1, incr=0)
reads, nbases = synth.make_reads(contigs, READS)
rec = 151
per = (int(os.environ.get("BATCH_MIB", "48")) << 20) // rec          # records per batch (the program hands over 32 MiB chunks)
with sk.KmerContext(0) as ctx:
    ctx.load_keyset(ks, 6)
    t = np.ones(ks.nrows, dtype=np.uint32)
    t[np.random.default_rng(3).choice(ks.nrows, ks.nrows // 100, replace=False)] = 2
    ctx.set_counts(0, t)
    starts = (np.arange(per, dtype=np.uint32) * rec)
    ctx.tally_batch(reads[: per * rec].tobytes(), starts, 0, 2)          # warm-up
    ctx.scan_timing(reset=True)
    t0 = time.perf_counter()
    hits = inf = 0
    for a in range(0, READS, per):
        n = min(per, READS - a)
        tally, h = ctx.tally_batch(reads[a * rec:(a + n) * rec].tobytes(), starts[:n], 0, 2)
        hits += int(tally[:, 0].sum())
        inf += int(tally[:, 1].sum())
        assert len(h) == int(tally[:, 1].sum())
    wall = time.perf_counter() - t0
    ms, launches = ctx.scan_timing(reset=True)
print(json.dumps({"mode": "tally (strain_detect)", "reads": READS, "bases": nbases, "kernel_ms_total": ms, "launches": launches,
                  "kernel_bases_per_s": nbases / (ms * 1e-3), "wall_bases_per_s_incl_h2d_d2h_sync": nbases / wall,
                  "window_hits": hits, "informative_hits": inf}))
