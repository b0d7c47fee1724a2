#!/usr/bin/env python3
"""tools/save_profile.py TAG -- copy the judged summaries of gpurun_out/prof_TAG into profiles/ and
refresh profiles/traffic.json (HBM bytes per launch from the PMC passes, gfx950 correction applied)."""
import glob
import hashlib
import json
import os
import shutil
import sys

tag = sys.argv[1]


def device_source_sha():
    """the same stamp as bench.py's device_source_sha(): sk_device.hip and its sk_dev_*.hip.h parts, names and bytes, in name order"""
    d = os.path.join("strainer2_amd", "csrc")
    h = hashlib.sha256()
    for name in ["sk_device.hip"] + sorted(n for n in os.listdir(d) if n.startswith("sk_dev_") and n.endswith(".hip.h")):
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


# the NEWEST run of the tag (tools/profile.sh writes every run into gpurun_out/prof_TAG_<time>; gpurun merges them all into this
# directory, so "the" run must be chosen, never globbed): by the directory's own time stamp in its name
runs = sorted(d for d in glob.glob(os.path.join("gpurun_out", "prof_" + tag + "_*")) if os.path.isfile(os.path.join(d, "summary.json")))
if not runs:
    sys.exit(f"no run of tag {tag} under gpurun_out/ (tools/profile.sh {tag} on the GPU box first)")
src = runs[-1]
s = json.load(open(os.path.join(src, "summary.json")))
want = "sk_scan_grid"
kname = [n for n in s["kernels"] if want in n]
if not kname:
    sys.exit(f"{src}: no {want} in the kernel trace")
# the kernel-stats file that is copied must be THIS run's: its average for the scan kernel is the one in summary.json
import csv
ks = os.path.join(src, "kernel_stats.csv")
rows = [r for r in csv.DictReader(open(ks)) if want in r["Name"]]
for r in rows:
    avg = s["kernels"][r["Name"].split("(")[0]]["avg_ns"]
    assert abs(float(r["AverageNs"]) - avg) < 1e-6 * max(avg, 1.0), f"{ks}: {r['Name'][:40]} averages {r['AverageNs']} ns, summary.json says {avg}"
assert rows, f"{ks}: no {want} row"
shutil.copy(os.path.join(src, "summary.txt"), f"profiles/{tag}_summary.txt")
shutil.copy(os.path.join(src, "summary.json"), f"profiles/{tag}_summary.json")
shutil.copy(ks, f"profiles/{tag}_kernel_stats.csv")
if len(sys.argv) > 2 and sys.argv[2] == "--no-traffic":          # (a profile of another kernel/workload: profiles/traffic.json stays)
    print(f"{tag}: copied from {src}")
    sys.exit(0)
c = [v for k, v in s["counters"].items() if "sk_scan_grid" in k][0]
fetch, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
bl = s.get("bench_line", {})
reads = bl.get("config", {}).get("reads_per_gpu", 10000000)
stream_bytes = reads * 151                       # the record stream: the only wide coalesced streaming read of the kernel
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B requests of a wide coalesced streaming read at
# 64 B, i.e. reports half of those bytes -- the missing half is added for the STREAMING share only (the record stream,
# read exactly once: stream_bytes / 2); the random 8/16-byte lookups (filter blocks, table slots, text, rank map) stay as counted
t = {"kernel": [n for n in s["kernels"] if "sk_scan_grid" in n][0].split("<")[0].replace("void ", ""), "reads": reads, "read_len": 150,
     "hbm_bytes_per_launch": fetch + stream_bytes / 2 + write, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
     "streaming_share_bytes": stream_bytes,
     "vector_instructions_per_launch": c.get("SQ_INSTS_VALU"), "gpu_cycles_per_launch_per_xcd": (c.get("GRBM_GUI_ACTIVE") or 0) / 8 or None,
     "tcc_requests_per_launch": c.get("TCC_REQ_sum"),
     "sk_device_hip_sha256": device_source_sha(),
     "correction": "FETCH_SIZE raw + half of the record stream's bytes (gfx950: 128-B streaming requests are tallied at 64 B; "
                   "applied to the streaming share only, random lookups as counted) + WRITE_SIZE (exact). Separate --pmc passes (tools/profile.sh).",
     "source": f"profiles/{tag}_summary.txt", "run": src}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
k = [v for n, v in s["kernels"].items() if "sk_scan_grid" in n][0]
print(f"{tag}: kernel-trace avg {k['avg_ns'] / 1e6:.3f} ms over {k['calls']} calls; bench events avg "
      f"{bl.get('roofline', {}).get('avg_launch_ms')} ms; traffic {t['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch")
