#!/usr/bin/env python3
"""Device time of one batch against NS resident strains: ONE scan against their union table (sk_union_*) against NS scans,
one per strain's table (sk_tally_launch on NS streams, as strain_detect -S did before the union).  The batch is resident
(filled once); wall clock over REPEAT launch + collect rounds, so PCIe upload is left out and the small result copies are
in.  Prints one JSON line (for DESIGN.md / profiles/, not the driver's bench)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import strainer2_amd as sk  # noqa: E402
from strainer2_amd import synth  # noqa: E402
from strainer2_amd.native import lib  # noqa: E402

NS = int(os.environ.get("NSTRAINS", "32"))
STRAIN_BP = int(os.environ.get("STRAIN_BP", "5000000"))
BATCH_MIB = int(os.environ.get("BATCH_MIB", "32"))
REPEAT = int(os.environ.get("REPEAT", "10"))
HIT = float(os.environ.get("HIT_FRAC", "0.02"))
rng = np.random.default_rng(5)
strains, ctxs, sets = [], [], []
t0 = time.time()
for s in range(NS):
    g = synth._rand_bases(rng, STRAIN_BP)
    strains.append(g)
    ks = sk.Keyset.from_stream(g.tobytes() + b"\n", default_val=1, incr=0)
    c = sk.KmerContext(0)
    c.load_keyset(ks, 6)
    typ = np.ones(ks.nrows, dtype=np.uint32)
    typ[rng.choice(ks.nrows, ks.nrows // 100, replace=False)] = 2
    c.set_counts(0, typ)
    ctxs.append(c)
    sets.append(ks)
print(f"{NS} strains resident in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
rec = 151
nrec = (BATCH_MIB << 20) // rec
blk = synth._rand_bases(rng, nrec * 150).reshape(nrec, 150)
for i in np.flatnonzero(rng.random(nrec) < HIT):
    g = strains[int(rng.integers(0, NS))]
    p0 = int(rng.integers(0, STRAIN_BP - 150))
    blk[i] = g[p0:p0 + 150]
stream = np.empty((nrec, rec), dtype=np.uint8)
stream[:, :150] = blk
stream[:, 150] = 10
stream = stream.tobytes()
starts = (np.arange(nrec, dtype=np.uint32) * rec)
nbases = nrec * 150

t0 = time.time()
u = sk.KmerUnion(ctxs, 0, 2)
t_build = time.time() - t0
tally_u, hits_u = u.tally_batch(stream, starts)              # fills the union's batch; warm-up; the answer
cap = max(len(hits_u) * 2, 1 << 16)
recs = np.zeros((nrec * NS + 1, 3), dtype=np.uint32)
hits = np.zeros((cap, 2), dtype=np.uint32)
nr, nh = C.c_uint64(0), C.c_uint64(0)
t0 = time.perf_counter()
for _ in range(REPEAT):
    assert lib.sk_union_tally_launch(u._h, u._batch, cap) == 0
    assert lib.sk_union_tally_collect(u._h, recs.ctypes.data, nrec * NS, C.byref(nr), hits.ctypes.data, C.byref(nh)) == 0
t_union = (time.perf_counter() - t0) / REPEAT
pairs_u, nh_u = nr.value, nh.value

one = np.zeros((nrec + 1, 3), dtype=np.uint32)
t0 = time.perf_counter()
pairs_m = nh_m = 0
for r in range(REPEAT + 1):
    if r == 1:
        t0 = time.perf_counter()
        pairs_m = nh_m = 0
    for c in ctxs:
        assert lib.sk_tally_launch(c._h, u._batch, 0, 2, cap) == 0
    for c in ctxs:
        assert lib.sk_tally_collect_sparse(c._h, one.ctypes.data, nrec, C.byref(nr), hits.ctypes.data, C.byref(nh)) == 0
        pairs_m += nr.value
        nh_m += nh.value
t_members = (time.perf_counter() - t0) / REPEAT
assert pairs_m == pairs_u * REPEAT and nh_m == nh_u * REPEAT, (pairs_m, pairs_u, nh_m, nh_u)
print(json.dumps({"strains": NS, "strain_bp": STRAIN_BP, "batch_mib": BATCH_MIB, "reads_in_batch": nrec, "strain_read_fraction": HIT,
                  "union_rows": u.rows, "union_build_s": round(t_build, 3),
                  "union_ms_per_batch": round(t_union * 1e3, 3), "members_ms_per_batch": round(t_members * 1e3, 3),
                  "union_gbase_per_s": round(nbases / t_union / 1e9, 1), "members_gbase_per_s_each_strain_counted_once": round(nbases / t_members / 1e9, 1),
                  "strain_x_gbase_per_s_union": round(NS * nbases / t_union / 1e9, 1), "strain_x_gbase_per_s_members": round(NS * nbases / t_members / 1e9, 1),
                  "read_strain_pairs_with_hits": pairs_u, "informative_hits": nh_u, "results_equal": True}))
u.close()
