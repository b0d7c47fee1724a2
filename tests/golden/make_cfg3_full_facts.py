#!/usr/bin/env python3
"""tests/golden/make_cfg3_full_facts.py -- pin BASELINE configs[2] AT SPEC (-A, -B and -C together) to the unmodified reference.

The job (strainer2_amd/cfg3.py): the 5 Mbp strain against a 1000-genome -A list (5 Gbase; ten of the genomes are the strain at
1 % divergence), a -B list of 67 x 1 M-read FASTQ files listed ten times over (10.05 Gbase distinct, 100.5 Gbase scanned) and a
-C list of five genomes that names the -r path itself, with a progress file.

The UNMODIFIED reference program (oracle/_ref/kmer_scrub_count, built by oracle/Makefile from /root/reference/src) runs here as P
processes over slices of the -A and -B lists (process 0 also takes -C and -p); the counters only ever get +1 per window
(src/genome_compare.c:220-223), so the slices' columns add up to the one-process result, and a list line that is listed ten
times is scanned ten times (src/genome_compare.c:163-172): the -B column of ONE pass over the 67 files times ten (mod 2^32) is
the column of the 670-line list.  From the slices' tables the script assembles the TSV the reference would print for the whole
job (same rows in the same order in every slice; reference_count identical) and records its md5 and line count, per column the
sum / non-zero rows / max / md5 of the u32 vector in the reference's row order, process 0's stderr (the skip message) and the
md5 of the progress file without its time stamps.

(The assembly was checked against the real thing at a size one process finishes: `--genomes 12 --b-files 3 --reads-per-file 20000`
gives stdout md5 b2488527...; the reference run ONCE on that whole job, its -B list naming the three files ten times over,
printed a table with the same md5 and the same progress file -- round 3, build container.)

tools/cfg3_full.py writes the same inputs on the GPU box, runs bin/kmer_scrub_count ONCE on the whole job and compares.
Only data is committed (tests/golden/cfg3_full_facts.json).  Runs in the build container only (about 15 minutes on 8 cores,
26 GB under --work).

  python3 tests/golden/make_cfg3_full_facts.py [--procs 8] [--work /tmp/cfg3_full]
"""
import argparse
import hashlib
import io
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from strainer2_amd import cfg3  # noqa: E402

EXE = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")
COLS = ["reference_count", "pangenome_count", "metagenome_count", "drug_count"]


def col_facts(v):
    v = v.astype(np.uint32)
    return {"sum": int(v.astype(np.uint64).sum()), "nonzero_rows": int(np.count_nonzero(v)), "max": int(v.max()),
            "md5_u32_le": hashlib.md5(v.astype("<u4").tobytes()).hexdigest()}


def tsv_md5(keys, cols):
    """md5 and length of the TSV the reference prints (src/kmer_scrub_count.c:134-156): constant 5-name header, %d of the unsigned counters"""
    h = hashlib.md5()
    head = b"#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n"
    h.update(head)
    n = len(head)
    step = 500_000
    signed = [c.astype(np.uint32).view(np.int32) for c in cols]
    for a in range(0, len(keys), step):
        buf = io.BytesIO()
        ks = keys[a:a + step]
        cs = [s[a:a + step] for s in signed]
        buf.write("".join(f"{k}\t" + "\t".join(str(int(c[i])) for c in cs) + "\n" for i, k in enumerate(ks)).encode())
        b = buf.getvalue()
        h.update(b)
        n += len(b)
    return h.hexdigest(), n


def progress_expected_md5(d):
    """what the reference's progress file holds without the times: the header, then every list line in turn -- also the skipped one
    (src/kmer_scrub_count.c:78-85, src/genome_compare.c:133-136,167-170)"""
    lines = [b"adding kmer counts for:"]
    for lst in ("A.txt", "B.txt", "C.txt"):
        lines += [ln for ln in open(os.path.join(d, lst), "rb").read().split(b"\n") if ln]
    return hashlib.md5(b"\n".join(lines + [b""])).hexdigest()


def main():
    import pandas as pd
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--work", default="/tmp/cfg3_full")
    ap.add_argument("--genomes", type=int, default=cfg3.N_GENOMES)
    ap.add_argument("--b-files", type=int, default=cfg3.N_B_FILES)
    ap.add_argument("--reads-per-file", type=int, default=cfg3.READS_PER_FILE)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "cfg3_full_facts.json"))
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    assert os.access(EXE, os.X_OK), "build oracle/_ref first (make -C oracle)"
    d = args.work
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    argv = cfg3.write_all(d, procs=args.procs, n_genomes=args.genomes, n_b=args.b_files, reads_per_file=args.reads_per_file,
                          progress=lambda n, m: print(f"  inputs {n}/{m} {time.time() - t0:.0f} s", flush=True))
    print(f"inputs written in {time.time() - t0:.0f} s", flush=True)
    P = args.procs
    a_lines = [cfg3.genome_name(i) for i in range(args.genomes)]
    b_lines = [cfg3.reads_name(j) for j in range(args.b_files)]             # ONE pass; the list names them LIST_REPEAT times
    # deal the slow items (the ten strain copies: every window is a hit) round-robin, like everything else
    ps = []
    t1 = time.time()
    for k in range(P):
        with open(os.path.join(d, f"A_{k}.txt"), "w") as f:
            f.write("".join(x + "\n" for x in a_lines[k::P]))
        with open(os.path.join(d, f"B_{k}.txt"), "w") as f:
            f.write("".join(x + "\n" for x in b_lines[k::P]))
        cmd = [EXE, "-r", "strain.fa", "-A", f"A_{k}.txt", "-B", f"B_{k}.txt"]
        if k == 0:
            cmd += ["-C", "C.txt", "-p", "progress_0.txt"]
        out = open(os.path.join(d, f"out_{k}.tsv"), "wb")
        err = open(os.path.join(d, f"err_{k}.txt"), "wb")
        ps.append((subprocess.Popen(cmd, cwd=d, stdout=out, stderr=err), out, err))
    for p, out, err in ps:
        assert p.wait() == 0, "reference program failed"
        out.close()
        err.close()
    print(f"reference: {P} processes in {time.time() - t1:.0f} s", flush=True)
    keys = None
    total = None
    for k in range(P):
        df = pd.read_csv(os.path.join(d, f"out_{k}.tsv"), sep="\t", header=None, skiprows=1, engine="c",
                         names=["kmer"] + COLS[:4 if k == 0 else 3], dtype={"kmer": str})
        cols = [(df[c].to_numpy(dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32) for c in COLS[:4 if k == 0 else 3]]
        if keys is None:
            keys = df["kmer"].tolist()
            total = [cols[0], cols[1].copy(), cols[2].copy(), cols[3].copy()]
        else:
            assert df["kmer"].tolist() == keys and np.array_equal(cols[0], total[0]), "slices disagree on rows"
            total[1] = (total[1] + cols[1]).astype(np.uint32)
            total[2] = (total[2] + cols[2]).astype(np.uint32)
        del df
    one_pass_meta = total[2].copy()
    total[2] = (total[2].astype(np.uint64) * cfg3.LIST_REPEAT & 0xFFFFFFFF).astype(np.uint32)
    md5, nbytes = tsv_md5(keys, total)
    facts = {
        "workload": "strainer2_amd/cfg3.py: 5 Mbp strain; -A %d genomes x 5 Mbp (10 strain copies at 1 %% divergence); -B %d FASTQ files x %d reads x %d bp "
                    "listed %d times; -C 5 genomes incl. the -r path; -p" % (args.genomes, args.b_files, args.reads_per_file, cfg3.READ_LEN, cfg3.LIST_REPEAT),
        "producer": "oracle/_ref/kmer_scrub_count (unmodified reference), %d processes over slices of the -A and -B lists (one pass over the -B files; "
                    "the list repeats them %d times: column x %d mod 2^32), columns added, the whole job's TSV assembled from the slices" % (P, cfg3.LIST_REPEAT, cfg3.LIST_REPEAT),
        "argv": argv, "genomes": args.genomes, "b_files": args.b_files, "reads_per_file": args.reads_per_file, "list_repeat": cfg3.LIST_REPEAT,
        "bases_scanned": {"A": args.genomes * 5_000_000, "B": args.b_files * args.reads_per_file * cfg3.READ_LEN * cfg3.LIST_REPEAT, "C": 4 * 5_000_000},
        "rows": len(keys), "stdout_md5": md5, "stdout_bytes": nbytes, "stdout_lines": len(keys) + 1,
        "columns": {c: col_facts(v) for c, v in zip(COLS, total)},
        "metagenome_count_one_pass": col_facts(one_pass_meta),
        "stderr": open(os.path.join(d, "err_0.txt")).read(),
        "md5_progress_without_times": progress_expected_md5(d),
        "first_rows": [keys[i] + "\t" + "\t".join(str(int(c.view(np.int32)[i])) for c in total) for i in range(3)],
    }
    with open(args.out, "w") as f:
        json.dump(facts, f, indent=1)
    print(json.dumps(facts)[:1500], flush=True)
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
