#!/usr/bin/env python3
"""tools/sigprof_report.py SAMPLES [top] -- samples of tools/probes/sigprof_preload.c as a table: share of the process's CPU time by
thread name, and inside every thread name by function (addr2line on the modules of this tree: the same binaries that ran)."""
import collections
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [ln.rstrip("\n").split("\t") for ln in open(sys.argv[1]) if ln.count("\t") == 2]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
by_mod = collections.defaultdict(set)
for name, mod, off in rows:
    by_mod[mod].add(off)
sym = {}
for mod, offs in by_mod.items():
    local = mod
    if "strainer2_amd/" in mod:
        local = os.path.join(REPO, mod[mod.index("strainer2_amd/"):])
    if not os.path.isfile(local) or "strainer2_amd/" not in mod and "/tmp/" not in mod and "build_exp" not in mod:
        for o in offs:                                       # (a system library without debug information: 256-byte buckets of its text)
            sym[(mod, o)] = "%s+0x%x00" % (os.path.basename(mod), int(o, 16) >> 8)
        continue
    offs = sorted(offs)
    out = subprocess.run(["addr2line", "-f", "-C", "-e", local] + ["0x" + o for o in offs], capture_output=True, text=True).stdout.split("\n")
    for i, o in enumerate(offs):
        fn = out[2 * i] if 2 * i < len(out) else "??"
        sym[(mod, o)] = (fn if fn != "??" else os.path.basename(mod) + "+?")
total = len(rows)
threads = collections.Counter(r[0] for r in rows)
print(f"{total} samples")
for name, n in threads.most_common():
    print(f"\n== {name}: {100.0 * n / total:.1f} % of the CPU time ({n} samples)")
    fns = collections.Counter(sym[(m, o)] for t, m, o in rows if t == name)
    for fn, k in fns.most_common(top):
        print(f"   {100.0 * k / n:5.1f} %  {fn}")
