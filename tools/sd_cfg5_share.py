#!/usr/bin/env python3
"""tools/sd_cfg5_share.py -- one GPU's share of BASELINE configs[4] at size: `strain_detect -S` with 32 strains of 5 Mbp resident
against a 100 Gbase SE metagenome (the 10 Gbase reads file of strainer2_amd/cfg5.py listed ten times in -B), on the GPU box.

  1. writes the inputs under WORK (default /dev/shm/sk_cfg5: 10.3 GB of FASTA + 32 strains + their -a lists)
  2. PIN: runs bin/strain_detect -S (all 32 strains, one union table) on the 1 Gbase PREFIX of the metagenome and compares the -o files of
     the two pinned strains (decompressed md5, lines, bytes, stdout, stderr) with tests/golden/cfg5_share_facts.json -- what the UNMODIFIED
     reference program wrote for exactly these inputs in the build container (tests/golden/make_cfg5_share_facts.py)
  3. SIZE: runs bin/strain_detect -S on the whole -B list (100 Gbase scanned), wall clock and the program's own timing lines; the
     size-independent checks: every strain's file must be LIST_REPEAT identical parts (one per list line: hit lines + the four
     trailer lines, src/strain_detect.c:263-384,633-636), and every part must BEGIN with the hit lines of that strain's prefix run
     (same reads in the same order; only the file name in column 1 differs)
Prints one JSON object (kept as profiles/r03_cfg5_share.json); exit status 1 on any difference.

  python3 tools/sd_cfg5_share.py        (env: WORK, PROCS, KEEP=1, SKIP_FULL=1; ONLY_PREFIX=1: write and run only the pinned prefix --
                                        32 strains resident x 1 Gbase, what tests/test_scale_checks_gpu.py runs)
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import cfg5  # noqa: E402

WORK = os.environ.get("WORK", "/dev/shm/sk_cfg5")
FACTS = os.path.join(REPO, "tests", "golden", "cfg5_share_facts.json")


def run(exe, argv, **env):
    prefix = env.pop("_PREFIX", "").split()                  # (a variant may put a launcher in front: "taskset -c 0-63,128-191")
    t = time.time()
    p = subprocess.run(prefix + [exe] + argv, cwd=WORK, capture_output=True, env=dict(os.environ, SK_SD_TIMING="1", **env))
    return p, time.time() - t


def split_timing(err):
    lines = err.decode(errors="replace").split("\n")
    timing = [ln for ln in lines if ln.startswith("strain_detect timing") or ln.startswith("strain_detect: no union table")]
    rest = "".join(ln + "\n" for ln in lines if ln and ln not in timing)
    return timing, rest


def main():
    facts = json.load(open(FACTS))
    assert facts["prefix_reads"] == cfg5.PREFIX_READS
    exe = os.path.join(REPO, "strainer2_amd", "bin", "strain_detect")
    t0 = time.time()
    only_prefix = bool(os.environ.get("ONLY_PREFIX"))
    paths = cfg5.write_all(WORK, procs=int(os.environ.get("PROCS", "16")), only_prefix=only_prefix,
                           progress=lambda n, m: print(f"  inputs {n}/{m} {time.time() - t0:.0f} s", file=sys.stderr, flush=True))
    t_write = time.time() - t0
    report = {"job": "strain_detect -S, %d strains x %d bp resident; SE FASTA of %d x %d bp reads (%.2f Gbase) listed %d times = %.1f Gbase scanned"
                     % (cfg5.NSTRAINS, cfg5.STRAIN_BP, cfg5.READS, cfg5.READ_LEN, cfg5.READS * cfg5.READ_LEN / 1e9, cfg5.LIST_REPEAT,
                        cfg5.LIST_REPEAT * cfg5.READS * cfg5.READ_LEN / 1e9),
              "inputs_written_in_s": round(t_write, 1)}
    ok = True
    # ---- 2. the prefix, pinned to the reference
    p, wall = run(exe, ["-S", paths["strains_prefix"], "-b", paths["prefix"], "-t", "SE"])
    timing, rest = split_timing(p.stderr)
    pin = {"wall_s": round(wall, 2), "returncode": p.returncode, "timing": timing, "strains": {}}
    for s, want in facts["strains"].items():
        hits = gzip.open(os.path.join(WORK, f"prefix{s}.gz"), "rb").read()
        got = {"hits_md5": hashlib.md5(hits).hexdigest(), "hits_bytes": len(hits), "hits_lines": hits.count(b"\n")}
        same = all(got[k] == want[k] for k in got) and p.returncode == want["returncode"] and p.stdout.decode() == want["stdout"] and rest == want["stderr"]
        pin["strains"][s] = dict(got, identical_to_the_reference=same)
        ok = ok and same
    report["prefix_%.1f_gbase_pinned" % (cfg5.PREFIX_READS * cfg5.READ_LEN / 1e9)] = pin
    report["facts"] = "tests/golden/cfg5_share_facts.json (" + facts["producer"] + ")"
    # ---- 3. the whole list
    if not os.environ.get("SKIP_FULL") and not only_prefix:
        p, wall = run(exe, ["-S", paths["strains"], "-B", paths["B"]])
        timing, rest = split_timing(p.stderr)
        bases = cfg5.LIST_REPEAT * cfg5.READS * cfg5.READ_LEN
        full = {"wall_s": round(wall, 2), "returncode": p.returncode, "bases_scanned": bases, "strain_x_bases_per_s": round(cfg5.NSTRAINS * bases / wall),
                "metagenome_bases_per_s": round(bases / wall), "timing": timing, "stderr_other": rest[-500:]}
        # size-independent check: a list line is scanned as often as it is listed, so every strain's file is LIST_REPEAT equal parts, and
        # a part begins with the hit lines of the prefix run (same reads, same order; the file name in column 1 differs)
        parts_ok, prefix_ok, lines = True, True, 0
        for s in range(cfg5.NSTRAINS):
            data = gzip.open(os.path.join(WORK, f"multi{s}.gz"), "rb").read()
            lines += data.count(b"\n")
            n = len(data) // cfg5.LIST_REPEAT
            part = data[:n]
            parts_ok = parts_ok and len(data) == n * cfg5.LIST_REPEAT and all(data[i * n:(i + 1) * n] == part for i in range(cfg5.LIST_REPEAT))
            pre = [ln.split(b"\t", 1)[1] for ln in gzip.open(os.path.join(WORK, f"prefix{s}.gz"), "rb").read().split(b"\n") if ln and not ln.startswith(b"#")]
            head = [ln.split(b"\t", 1)[1] for ln in part.split(b"\n")[:len(pre)]]
            prefix_ok = prefix_ok and head == pre
        full["every_file_is_%d_equal_parts" % cfg5.LIST_REPEAT] = parts_ok
        full["every_part_begins_with_the_prefix_runs_hit_lines"] = prefix_ok
        full["output_lines"] = lines
        ok = ok and parts_ok and prefix_ok and p.returncode == 0
        report["full_pass"] = full
        # A/B of the same pass (VARIANTS=1): without the scan-ahead of the next chunk, with bigger chunks, on several logical devices
        if os.environ.get("VARIANTS"):
            def two_md5():                                   # (decompressed: the gz members differ with the block boundaries)
                return [hashlib.md5(gzip.open(os.path.join(WORK, f"multi{s}.gz"), "rb").read()).hexdigest() for s in cfg5.PINNED_STRAINS]
            sizes = two_md5()
            report["variants"] = {}
            for name, env in (("no_scan_ahead", {"SK_SD_NO_AHEAD": "1"}), ("chunks_of_128_mib", {"SK_SD_CHUNK_BYTES": str(128 << 20)}),
                              ("parse_threads_2", {"SK_PARSE_THREADS": "2"}), ("parse_threads_6", {"SK_PARSE_THREADS": "6"}),
                              ("parse_threads_8", {"SK_PARSE_THREADS": "8"}),
                              ("parse_threads_8_chunks_of_128_mib", {"SK_PARSE_THREADS": "8", "SK_SD_CHUNK_BYTES": str(128 << 20)}),
                              ("default_again", {}), ("chunks_packed", {"SK_SD_PACK": "1"}), ("chunks_packed_again", {"SK_SD_PACK": "1"}), ("chunks_packed_16_parsers", {"SK_SD_PACK": "1", "SK_PARSE_THREADS": "16"}), ("sync_blocking", {"SK_SYNC": "blocking"}), ("sync_yield", {"SK_SYNC": "yield"}), ("sync_blocking_again", {"SK_SYNC": "blocking"}), ("input_pread", {"SK_SD_INPUT": "pread"}), ("input_mapped", {"SK_SD_INPUT": "mapped"}),
                              ("default_once_more", {}), ("input_pread_again", {"SK_SD_INPUT": "pread"}),
                              # (sampled: gcc -O2 -shared -fPIC -o /tmp/sigprof.so tools/probes/sigprof_preload.c -ldl first; tools/sigprof_report.py reads the samples)
                              ("sigprof", {"LD_PRELOAD": "/tmp/sigprof.so", "SK_LEAK_AT_EXIT": "0", "SK_PROF_OUT": os.path.join(REPO, "gpurun_out", "sigprof_sd.txt")}),
                              ("cpus_of_node_0", {"_PREFIX": "taskset -c 0-63,128-191"}), ("cpus_of_node_1", {"_PREFIX": "taskset -c 64-127,192-255"}),
                              ("cpus_of_node_0_again", {"_PREFIX": "taskset -c 0-63,128-191"}), ("cpus_of_node_1_again", {"_PREFIX": "taskset -c 64-127,192-255"}),
                              ("read_block_512k", {"SK_SD_INPUT": "pread", "SK_READ_BLOCK": str(512 << 10)}), ("read_block_1m", {"SK_SD_INPUT": "pread", "SK_READ_BLOCK": str(1 << 20)}),
                              ("read_block_4m", {"SK_SD_INPUT": "pread", "SK_READ_BLOCK": str(4 << 20)}), ("read_block_32m", {"SK_SD_INPUT": "pread", "SK_READ_BLOCK": str(32 << 20)}),
                              ("two_logical_devices_one_card", {"SK_DEVICES": "0,0", "SK_SD_GROUP": "16"})):
                if os.environ.get("VARIANTS") not in ("1", "all") and name not in os.environ["VARIANTS"].split(","):
                    continue
                p2, wall2 = run(exe, ["-S", paths["strains"], "-B", paths["B"]], **env)
                timing2, _ = split_timing(p2.stderr)
                same = p2.returncode == 0 and sizes == two_md5()
                report["variants"][name] = {"env": env, "wall_s": round(wall2, 2), "same_output_for_the_two_pinned_strains": same, "timing": timing2}
                ok = ok and same
    report["ok"] = ok
    print(json.dumps(report, indent=1))
    if not os.environ.get("KEEP"):
        shutil.rmtree(WORK, ignore_errors=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
