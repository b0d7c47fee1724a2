#!/usr/bin/env python3
"""tools/exp_grid.py -- one process, many variants of the scan kernel (options, ablations, strain-read fractions).

Prints, per variant, the average launch time of the scan kernel (HIP events on the library's stream).  Under
`rocprofv3 --kernel-trace --pmc ...` the launches come out in the order of the variants, LAUNCHES per variant
(tools/exp_grid_pmc.py assigns them).  Run on the GPU box.

  python3 tools/exp_grid.py [--reads N] [--launches L] [--variants name,name,...]
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# name -> (hit_frac, options, uncached stream[, substitution rate of the strain reads (default 0.005)])
VARIANTS = {
    "base_h0":      (0.0,  {}, False),
    "base_h2":      (0.02, {}, False),
    "nol2_h0":      (0.0,  {"ablate": 5}, False),
    "nol2_h2":      (0.02, {"ablate": 5}, False),
    "l2hit_h2":     (0.02, {"ablate": 4}, False),
    "l1hit_h2":     (0.02, {"ablate": 6}, False),
    "nofilt_h2":    (0.02, {"ablate": 1}, False),
    "l1only_h0":    (0.0,  {"ablate": 10}, False),
    "l1only_h2":    (0.02, {"ablate": 10}, False),
    "g2048_h0":     (0.0,  {"grid_kib": 2048}, False),
    "g2048_h2":     (0.02, {"grid_kib": 2048}, False),
    "g1536_h2":     (0.02, {"grid_kib": 1536}, False),
    "g4096_h2":     (0.02, {"grid_kib": 4096}, False),
    "g8192_h2":     (0.02, {"grid_kib": 8192}, False),
    **{f"g{k}_h{h}": (f, {"grid_kib": k}, False) for k in (3072, 3584, 4096, 4608, 5120, 6144, 8192, 12288) for h, f in ((0, 0.0), (2, 0.02), (30, 0.3)) if f"g{k}_h{h}" not in ("g4096_h2", "g8192_h2")},
    "unc_h0":       (0.0,  {}, True),
    "unc_h2":       (0.02, {}, True),
    "unc_g2048_h2": (0.02, {"grid_kib": 2048}, True),
    "notext_h2":    (0.02, {"text_stage": 0}, False),
    "notext_h100":  (1.0,  {"text_stage": 0}, False),
    "pipe7_h2":     (0.02, {"ablate": 7}, False),
    "pipe8_h2":     (0.02, {"ablate": 8}, False),
    "pipe9_h2":     (0.02, {"ablate": 9}, False),
    "abl8_h2":      (0.02, {"ablate": 8}, False),
    "abl9_h2":      (0.02, {"ablate": 9}, False),
    "single_h0":    (0.0,  {"pipeline": 1}, False),
    "single_h2":    (0.02, {"pipeline": 1}, False),
    "single_h30":   (0.3,  {"pipeline": 1}, False),
    "single_h100":  (1.0,  {"pipeline": 1}, False),
    "exact_h100":   (1.0,  {}, False, 0.0),
    "exact_h30":    (0.3,  {}, False, 0.0),
    "exact_noprobe_h30": (0.3, {"ablate": 2}, False, 0.0),
    "noatom_h30":   (0.3,  {"ablate": 3}, False),
    "noprobe_h30":  (0.3,  {"ablate": 2}, False),
    "notext_h30":   (0.3,  {"text_stage": 0}, False),
    "div1_h100":    (1.0,  {}, False, 0.01),
    "div3_h100":    (1.0,  {}, False, 0.03),
    "base_h30":     (0.3,  {}, False),
    "base_h100":    (1.0,  {}, False),
    # where a launch of the dense regimes goes (round 4: both run at ~3.2 TB/s of L2 misses): no counter atomics, no table probes, smaller filter
    "noatom_h100":  (1.0,  {"ablate": 3}, False),
    "noprobe_h100": (1.0,  {"ablate": 2}, False),
    "g2048_h100":   (1.0,  {"grid_kib": 2048}, False),
    "g1024_h100":   (1.0,  {"grid_kib": 1024}, False),
    "noatom_div3":  (1.0,  {"ablate": 3}, False, 0.03),
    "noprobe_div3": (1.0,  {"ablate": 2}, False, 0.03),
    "g2048_div3":   (1.0,  {"grid_kib": 2048}, False, 0.03),
    "g1024_div3":   (1.0,  {"grid_kib": 1024}, False, 0.03),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed launches of the variant for this long before its timed ones (the card's clocks ramp; the host-side set-up between "
                         "two variants lets them fall again).  0 under rocprofv3 --pmc, where the launches are told apart by their order")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    args = ap.parse_args()
    names = [v for v in args.variants.split(",") if v]
    from strainer2_amd import synth
    import strainer2_amd as sk

    contigs = synth.make_strain()
    sstream = synth.strain_stream(contigs)
    ks = sk.Keyset.from_stream(sstream)
    reads_by_frac = {}
    out = []
    for name in names:
        frac, opts, unc = VARIANTS[name][:3]
        sub = VARIANTS[name][3] if len(VARIANTS[name]) > 3 else 0.005
        if (frac, sub) not in reads_by_frac:
            reads_by_frac[(frac, sub)] = synth.make_reads(contigs, args.reads, 150, hit_frac=frac, sub_rate=sub, seed=synth.SEED + 1)
        reads, nbases = reads_by_frac[(frac, sub)]
        ctx = sk.KmerContext(0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        if unc:
            ctx.set_option("dev_alloc_uncached", 1)
        ctx.load_keyset(ks, 4)
        dev = ctx.dev_alloc(reads.size)
        ctx.dev_upload(dev, reads)
        if args.preheat_ms > 0:
            import time
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < args.preheat_ms:
                for _ in range(10):
                    ctx.scan_device(dev, int(reads.size), 2)
                ctx.sync()
            ctx.zero_counts(2)
        ctx.scan_device(dev, int(reads.size), 2)           # warm-up (counted as a launch of the variant)
        ctx.sync()
        ctx.scan_timing(reset=True)
        for _ in range(args.launches - 1):
            ctx.scan_device(dev, int(reads.size), 2)
        ctx.sync()
        ms, n = ctx.scan_timing(reset=True)
        hits = int(ctx.counts(2).astype(np.uint64).sum()) // args.launches
        rec = {"variant": name, "ms": ms / max(n, 1), "gbase_s": nbases / (ms / max(n, 1) * 1e-3) / 1e9, "hits_per_pass": hits}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        ctx.dev_free(dev)
        ctx.close()
    print("ORDER " + ",".join(names), flush=True)


if __name__ == "__main__":
    main()
