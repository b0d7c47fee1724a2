#!/usr/bin/env python3
"""tools/probes/phase_clock.py -- where a wave of the scan kernel spends its cycles (experiment build: tools/exp_variant_build.sh pc "-DSK_PHASE_CLOCK=1").
   SK_LIBRARY=$PWD/build_exp/libsk_pc.so python3 tools/probes/phase_clock.py [hit fractions...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import strainer2_amd as sk
from strainer2_amd import synth
from strainer2_amd.native import lib

NAMES = ["start -> loads issued", "wait for the loads + decode", "barrier", "records, hashes, lookups issued", "wait for the lookups + verdicts",
         "second and third questions", "stage 2", "closing barrier + flush"]
contigs = synth.make_strain()
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
for frac in [float(x) for x in (sys.argv[1:] or ["0", "0.02", "0.3"])]:
    reads, nb = synth.make_reads(contigs, 4_000_000, 150, hit_frac=frac, seed=synth.SEED + 1)
    with sk.KmerContext(0) as ctx:
        ctx.load_keyset(ks, 4)
        dev = ctx.dev_alloc(reads.size)
        ctx.dev_upload(dev, reads)
        ctx.scan_device(dev, int(reads.size), 2)
        out = (C.c_ulonglong * 8)()
        lib.sk_debug_phase_clock.argtypes = [C.c_void_p, C.c_void_p]
        lib.sk_debug_phase_clock(ctx._h, out)
        ctx.scan_timing(reset=True)
        for _ in range(4):
            ctx.scan_device(dev, int(reads.size), 2)
        ctx.sync()
        ms, n = ctx.scan_timing(reset=True)
        lib.sk_debug_phase_clock(ctx._h, out)
        tot = sum(out)
        print(f"strain reads {frac:.2f}: {ms / n:.4f} ms per launch; wave-time by phase (s_memtime ticks, % of the sum):")
        for k in range(8):
            print(f"   {NAMES[k]:36s} {out[k] / 4 / 1e6:10.2f} M  {100.0 * out[k] / tot:5.1f} %")
