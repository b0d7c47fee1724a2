#!/usr/bin/env python3
"""tests/golden/make_cfg2_facts.py -- pin the BASELINE-size result of configs[1] (the workload bench.py times).

Runs the UNMODIFIED reference program (oracle/_ref/kmer_scrub_count, built by oracle/Makefile from
/root/reference/src where that exists) on exactly the stream bench.py scans -- synth.make_strain() against
synth.make_reads(contigs, 10_000_000, 150, hit_frac=0.02, seed=SEED + 1 + rank) -- and records, per rank 0..7,
facts about the metagenome_count column in the reference's row order (src/kmer_scrub_count.c:134-156):
sum, number of non-zero rows, md5 of the little-endian u32 vector.  The reads are cut into P slices scanned by
P processes (the counters are commutative, src/genome_compare.c:220-223); the slices' columns are added.

bench.py asserts its counts against these facts inside the timed run; tests/test_gpu_parity.py does the same
through sk_scan_device.  Only data is committed (tests/golden/cfg2_facts.json); this script runs in the build
container only.

  python3 tests/golden/make_cfg2_facts.py [--ranks 8] [--procs 8] [--reads 10000000]
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402

EXE = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")


def column_of(tsv_path, ncol=3):
    """the metagenome_count column of a reference TSV (header + `kmer ref pan meta` rows), as u32"""
    import pandas as pd
    df = pd.read_csv(tsv_path, sep="\t", header=0, usecols=[ncol], dtype=np.int64, engine="c")
    return (df.iloc[:, 0].to_numpy() & 0xFFFFFFFF).astype(np.uint32)       # %d of an unsigned


def facts_for(contigs, work, reads, read_len, procs):
    rec = read_len + 1
    n = reads.size // rec
    per = (n + procs - 1) // procs
    head = np.frombuffer(b">r\n", dtype=np.uint8)
    lists = []
    for i in range(procs):
        rows = reads[i * per * rec:min(n, (i + 1) * per) * rec].reshape(-1, rec)
        fa = np.empty((rows.shape[0], 3 + rec), dtype=np.uint8)
        fa[:, :3] = head
        fa[:, 3:] = rows
        fa.tofile(os.path.join(work, f"reads{i}.fa"))
        with open(os.path.join(work, f"B{i}.txt"), "w") as f:
            f.write(os.path.join(work, f"reads{i}.fa") + "\n")
        lists.append(os.path.join(work, f"B{i}.txt"))
    ps = []
    for i, b in enumerate(lists):
        out = open(os.path.join(work, f"out{i}.tsv"), "wb")
        ps.append((subprocess.Popen([EXE, "-r", os.path.join(work, "strain.fa"), "-A", os.path.join(work, "empty.txt"), "-B", b],
                                    stdout=out, stderr=subprocess.DEVNULL), out))
    for p, out in ps:
        assert p.wait() == 0, "reference program failed"
        out.close()
    total = None
    for i in range(procs):
        col = column_of(os.path.join(work, f"out{i}.tsv"))
        total = col if total is None else (total + col).astype(np.uint32)
        os.remove(os.path.join(work, f"out{i}.tsv"))
        os.remove(os.path.join(work, f"reads{i}.fa"))
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "cfg2_facts.json"))
    args = ap.parse_args()
    assert os.access(EXE, os.X_OK), "build oracle/_ref first (make -C oracle)"
    contigs = synth.make_strain()
    work = tempfile.mkdtemp(prefix="cfg2_facts_")
    facts = {"workload": "synth.make_strain() vs synth.make_reads(contigs, %d, %d, hit_frac=0.02, seed=SEED+1+rank)" % (args.reads, args.read_len),
             "producer": "oracle/_ref/kmer_scrub_count (unmodified reference), %d processes over slices, columns added" % args.procs,
             "column": "metagenome_count, reference row order", "reads": args.reads, "read_len": args.read_len, "ranks": []}
    try:
        with open(os.path.join(work, "strain.fa"), "wb") as f:
            f.write(synth.strain_fasta(contigs))
        open(os.path.join(work, "empty.txt"), "w").close()
        for rank in range(args.ranks):
            t0 = time.time()
            reads, _ = synth.make_reads(contigs, args.reads, args.read_len, hit_frac=0.02, seed=synth.SEED + 1 + rank)
            col = facts_for(contigs, work, reads, args.read_len, args.procs)
            facts["ranks"].append({"rank": rank, "rows": int(col.size), "sum": int(col.astype(np.uint64).sum()),
                                   "nonzero_rows": int(np.count_nonzero(col)), "max": int(col.max()),
                                   "md5_u32_le": hashlib.md5(col.astype("<u4").tobytes()).hexdigest()})
            print(facts["ranks"][-1], f"{time.time() - t0:.0f} s", flush=True)
            with open(args.out, "w") as f:
                json.dump(facts, f, indent=1)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
