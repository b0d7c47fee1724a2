#!/usr/bin/env python3
"""tools/save_profile.py TAG -- copy the judged summaries of gpurun_out/prof_TAG into profiles/ and
refresh profiles/traffic.json (HBM bytes per launch from the PMC passes, gfx950 correction applied)."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
s = json.load(open(os.path.join(src, "summary.json")))
shutil.copy(os.path.join(src, "summary.txt"), f"profiles/{tag}_summary.txt")
shutil.copy(os.path.join(src, "summary.json"), f"profiles/{tag}_summary.json")
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
c = [v for k, v in s["counters"].items() if "sk_scan_grid" in k or "sk_scan_main" in k][0]
fetch, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
bl = s.get("bench_line", {})
t = {"kernel": [n for n in s["kernels"] if "sk_scan_grid" in n or "sk_scan_main" in n][0].split("<")[0].replace("void ", ""), "reads": bl.get("config", {}).get("reads_per_gpu", 10000000), "read_len": 150,
     "hbm_bytes_per_launch": 2 * fetch + write, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
     "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> doubled; "
                   "WRITE_SIZE exact. Separate --pmc passes (tools/profile.sh).",
     "source": f"profiles/{tag}_summary.txt"}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
k = [v for n, v in s["kernels"].items() if "sk_scan_grid" in n or "sk_scan_main" in n][0]
print(f"{tag}: kernel-trace avg {k['avg_ns'] / 1e6:.3f} ms over {k['calls']} calls; bench events avg "
      f"{bl.get('roofline', {}).get('avg_launch_ms')} ms; traffic {t['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch")
