import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import strainer2_amd as sk
from strainer2_amd import synth
READS = 10_000_000
contigs = synth.make_strain()
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
reads, nb = synth.make_reads(contigs, READS, 150, seed=synth.SEED + 1)
with sk.KmerContext(0) as ctx:
    ctx.load_keyset(ks, 4)
    dev = ctx.dev_alloc(reads.size)
    for label, edit in (("no odd byte", 0), ("one read in 1000 with an IUPAC letter", READS // 1000), ("one read in 20", READS // 20)):
        r = reads.copy()
        if edit:
            rng = np.random.default_rng(3)
            idx = rng.choice(READS, size=edit, replace=False).astype(np.int64) * 151 + 75
            r[idx] = ord("R")
        ctx.dev_upload(dev, r)
        ctx.scan_device(dev, int(r.size), 2); ctx.sync()
        t = time.perf_counter()
        for _ in range(20): ctx.scan_device(dev, int(r.size), 2)
        ctx.sync()
        print(label, round((time.perf_counter() - t) / 20 * 1e3, 3), "ms wall per pass")
