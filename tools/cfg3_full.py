#!/usr/bin/env python3
"""tools/cfg3_full.py -- BASELINE configs[2] AT SPEC as ONE job on the GPU box, against the unmodified reference's facts.

Writes the inputs of strainer2_amd/cfg3.py under WORK (default /dev/shm/sk_cfg3: 26 GB -- 1000 genomes x 5 Mbp for -A, 67 FASTQ
files x 1 M reads for -B listed ten times over = 100.5 Gbase scanned, 5 genomes for -C with the -r path among them), runs
bin/kmer_scrub_count ONCE on the whole job (-r -A -B -C -p) and compares with tests/golden/cfg3_full_facts.json, which the
UNMODIFIED reference program produced in the build container (tests/golden/make_cfg3_full_facts.py):
md5 + length of the TSV on stdout, per column sum / non-zero rows / max / md5 of the u32 vector in the reference's row order,
stderr, and the progress file without its time stamps.  Prints one JSON object (kept as profiles/r03_cfg3_full_facts.txt);
exit status 1 on any difference.

  python3 tools/cfg3_full.py            (env: WORK, PROCS = writers, SK_THREADS = decode threads of the program, KEEP=1,
                                        LIST_REPEAT=1: the -B files listed ONCE -- the facts hold the one-pass column of the
                                        reference's run (metagenome_count_one_pass; the ten-fold list is that column x 10 mod 2^32),
                                        so the job is as pinned as at x 10 and fits a test suite: 15.5 Gbase scanned.  What cannot be
                                        compared then is the md5 of the whole TSV and of the progress file, which name the ten-fold
                                        list; the progress file is checked line by line against the lists instead.)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import cfg3  # noqa: E402

WORK = os.environ.get("WORK", "/dev/shm/sk_cfg3")
FACTS = os.path.join(REPO, "tests", "golden", "cfg3_full_facts.json")
COLS = ["reference_count", "pangenome_count", "metagenome_count", "drug_count"]


def main():
    import pandas as pd
    facts = json.load(open(FACTS))
    procs = int(os.environ.get("PROCS", "16"))
    t0 = time.time()
    repeat = int(os.environ.get("LIST_REPEAT", str(facts["list_repeat"])))
    argv = cfg3.write_all(WORK, procs=procs, n_genomes=facts["genomes"], n_b=facts["b_files"], reads_per_file=facts["reads_per_file"], repeat=repeat,
                          progress=lambda n, m: print(f"  inputs {n}/{m} {time.time() - t0:.0f} s", file=sys.stderr, flush=True))
    t_write = time.time() - t0
    assert argv == facts["argv"]
    exe = os.path.join(REPO, "strainer2_amd", "bin", "kmer_scrub_count")
    out_path = os.path.join(WORK, "out.tsv")
    t1 = time.time()
    with open(out_path, "wb") as out:
        p = subprocess.run([exe] + argv, cwd=WORK, stdout=out, stderr=subprocess.PIPE, env=dict(os.environ, SK_TIMING="1"))
    wall = time.time() - t1
    is_timing = lambda ln: ln.startswith("kmer_scrub_count timing") or ln.startswith("key set of ")      # (SK_TIMING=1 lines) # noqa: E731
    timing = [ln for ln in p.stderr.decode().split("\n") if is_timing(ln)]
    stderr = "".join(ln + "\n" for ln in p.stderr.decode().split("\n") if ln and not is_timing(ln))
    h = hashlib.md5()
    nbytes = 0
    with open(out_path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
            nbytes += len(blk)
    df = pd.read_csv(out_path, sep="\t", header=None, skiprows=1, engine="c", names=["kmer"] + COLS, dtype={"kmer": str})
    got_cols = {}
    for c in COLS:
        v = (df[c].to_numpy(dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32)
        got_cols[c] = {"sum": int(v.astype(np.uint64).sum()), "nonzero_rows": int(np.count_nonzero(v)), "max": int(v.max()),
                       "md5_u32_le": hashlib.md5(v.astype("<u4").tobytes()).hexdigest()}
    prog = [ln.split(b"\t")[0] for ln in open(os.path.join(WORK, "progress.txt"), "rb").read().split(b"\n")]
    got = {"returncode": p.returncode, "stdout_md5": h.hexdigest(), "stdout_bytes": nbytes, "stdout_lines": len(df) + 1, "columns": got_cols,
           "stderr": stderr, "md5_progress_without_times": hashlib.md5(b"\n".join(prog)).hexdigest()}
    want = {k: facts[k] for k in ("stdout_md5", "stdout_bytes", "stdout_lines", "columns", "stderr", "md5_progress_without_times")}
    want["returncode"] = 0
    bases = sum(facts["bases_scanned"].values())
    if repeat != facts["list_repeat"]:
        assert repeat == 1, "the facts hold the ten-fold list and the one-pass column"
        want["columns"] = dict(facts["columns"], metagenome_count=facts["metagenome_count_one_pass"])
        for k in ("stdout_md5", "stdout_bytes", "md5_progress_without_times"):
            del want[k]
        # the progress file, line by line: the header, then every list line in order (-A, -B, -C; src/kmer_scrub_count.c:78-94)
        lists = [open(os.path.join(WORK, argv[argv.index(f) + 1])).read().split("\n")[:-1] for f in ("-A", "-B", "-C")]
        want["progress_lines"] = ["adding kmer counts for:"] + [ln for lst in lists for ln in lst] + [""]
        got["progress_lines"] = [ln.decode() for ln in prog]
        bases = facts["bases_scanned"]["A"] + facts["bases_scanned"]["B"] // facts["list_repeat"] + facts["bases_scanned"]["C"]
    diffs = [k for k in want if got[k] != want[k]]
    report = {"job": facts["workload"], "facts": "tests/golden/cfg3_full_facts.json (" + facts["producer"] + ")",
              "identical_to_the_reference": not diffs, "differences": diffs,
              "list_repeat": repeat, "bases_scanned_total": bases, "wall_seconds_program": round(wall, 2), "bases_per_s_end_to_end": round(bases / wall),
              "program_timing": timing, "decode_threads": os.environ.get("SK_THREADS", "default (CPU budget, at most 16)"),
              "inputs_written_in_s": round(t_write, 1), "got": {k: v for k, v in got.items() if k != "progress_lines"}}
    print(json.dumps(report, indent=1))
    if not os.environ.get("KEEP"):
        shutil.rmtree(WORK, ignore_errors=True)
    return 1 if diffs else 0


if __name__ == "__main__":
    sys.exit(main())
