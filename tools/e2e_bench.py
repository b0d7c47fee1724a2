#!/usr/bin/env python3
"""End-to-end (from files, through the drop-in program) rate: synthetic 5 Mbp strain, NFILES read
files of READS reads each as plain FASTQ and as .gz; wall clock of strainer2_amd/bin/kmer_scrub_count
with 1 and with SK_THREADS host decode threads.  Numbers go to DESIGN.md (never bench.py's `value`)."""
import gzip
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from strainer2_amd import synth  # noqa: E402

NFILES = int(os.environ.get("NFILES", "16"))
READS = int(os.environ.get("READS", "500000"))
work = os.environ.get("WORK", "/tmp/sk_e2e")
os.makedirs(work, exist_ok=True)
contigs = synth.make_strain(total_bp=int(os.environ.get("STRAIN_BP", "5000000")))
open(os.path.join(work, "strain.fa"), "wb").write(synth.strain_fasta(contigs))
qual = b"I" * 150
# QUAL=binned: quality strings drawn per base from four binned values (as current Illumina machines write them)
# instead of a constant line -- gzip then compresses ~4.5x instead of ~6x and inflate has real work to do
BINNED = os.environ.get("QUAL") == "binned"
t0 = time.time()
KINDS = tuple(os.environ.get("KINDS", "fq,fq.gz").split(","))
for kind in KINDS:
    names = []
    for i in range(NFILES):
        p = os.path.join(work, f"reads{i}.{kind}")
        names.append(p)
        if os.path.exists(p):
            continue
        stream, _ = synth.make_reads(contigs, READS, seed=synth.SEED + 100 + i)
        rows = stream.reshape(READS, 151)[:, :150]
        if BINNED:
            import numpy as np
            q = np.random.default_rng(1000 + i).choice(np.frombuffer(b"FFFFFFFFFF::,#", dtype=np.uint8), size=(READS, 150))
            body = b"".join(b"@r%d\n%s\n+\n%s\n" % (j, rows[j].tobytes(), q[j].tobytes()) for j in range(READS))
            del q
        else:
            body = b"".join(b"@r%d\n%s\n+\n%s\n" % (j, rows[j].tobytes(), qual) for j in range(READS))
        if kind.endswith("gz"):
            with gzip.open(p, "wb", compresslevel=4) as f:
                f.write(body)
        else:
            open(p, "wb").write(body)
        print(f"  wrote {p} ({time.time() - t0:.0f} s)", flush=True)
    open(os.path.join(work, f"B_{kind}.txt"), "w").write("\n".join(names) + "\n")
open(os.path.join(work, "A.txt"), "w").write(os.path.join(work, "strain.fa") + "\n")
print(f"inputs ready in {time.time() - t0:.1f} s: {NFILES} x {READS} reads per format", flush=True)
exe = os.path.join(REPO, "strainer2_amd", "bin", "kmer_scrub_count")
bases = NFILES * READS * 150
for kind in KINDS:
    for threads in tuple(int(x) for x in os.environ.get("THREADS", "1,4,16").split(",")):
        env = dict(os.environ, SK_THREADS=str(threads))
        t = time.time()
        with open(os.devnull, "wb") as null:
            subprocess.run([exe, "-r", os.path.join(work, "strain.fa"), "-A", os.path.join(work, "A.txt"),
                            "-B", os.path.join(work, f"B_{kind}.txt")], stdout=null, check=True, env=env)
        dt = time.time() - t
        print(f"{kind:6s} SK_THREADS={threads:2d}: {dt:6.2f} s wall for {bases / 1e9:.2f} Gbase in -B  "
              f"=> {bases / dt / 1e9:.3f} Gbase/s end to end (includes strain build, table print)", flush=True)

# REPEAT=N: the .gz list N times over in one -B (cfg 3 at full size without N times the disk space), and the
# size-independent parity property that goes with it: every metagenome_count must be N x the one-pass count
REPEAT = int(os.environ.get("REPEAT", "0"))
if REPEAT > 1:
    import pandas as pd
    kind = KINDS[-1]
    names = open(os.path.join(work, f"B_{kind}.txt")).read().split()
    open(os.path.join(work, "B_repeat.txt"), "w").write("\n".join(names * REPEAT) + "\n")
    env = dict(os.environ, SK_THREADS=os.environ.get("THREADS", "16").split(",")[-1])
    outs = []
    for lst, label in ((f"B_{kind}.txt", "one pass"), ("B_repeat.txt", f"{REPEAT} passes")):
        out = os.path.join(work, "counts_" + label.split()[0] + ".tsv")
        t = time.time()
        with open(out, "wb") as f:
            subprocess.run([exe, "-r", os.path.join(work, "strain.fa"), "-A", os.path.join(work, "A.txt"), "-B", os.path.join(work, lst)],
                           stdout=f, check=True, env=env)
        dt = time.time() - t
        n = bases * (REPEAT if lst == "B_repeat.txt" else 1)
        print(f"{label:10s} of the {kind} list, SK_THREADS={env['SK_THREADS']}: {dt:6.2f} s wall for {n / 1e9:.1f} Gbase => {n / dt / 1e9:.2f} Gbase/s end to end", flush=True)
        outs.append(pd.read_csv(out, sep="\t"))
    a, b = outs
    same_keys = (a["#kmer"] == b["#kmer"]).all() and (a["reference_count"] == b["reference_count"]).all() and (a["pangenome_count"] == b["pangenome_count"]).all()
    linear = (b["metagenome_count"] == REPEAT * a["metagenome_count"]).all()
    print(f"rows {len(a)}, hits in one pass {int(a['metagenome_count'].sum())}; same rows/other columns: {bool(same_keys)}; "
          f"metagenome_count of {REPEAT} passes == {REPEAT} x one pass in every row: {bool(linear)}", flush=True)
    assert same_keys and linear
