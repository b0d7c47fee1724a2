#!/usr/bin/env python3
"""tests/golden/make_cfg3_shape.py -- a BASELINE configs[2]-SHAPED job, and what the unmodified reference prints for it.

configs[2] = one strain against a 100 Gbp metagenome (-B), a 1000-genome -A list and a -C co-occurring set.  The shape, at
a size the reference finishes in under a minute: a 300 kbp strain (3 contigs, wrapped lines, a few N), an -A list of
1000 genome files of 20 kbp (10 of them copies of strain segments with 1 % divergence, half of those on the other
strand; the rest random), a -B list of 8 read files (plain and .gz FASTQ, 2 % of the reads cut from the strain with
0.5 % errors) and a -C list of 5 genomes that contains the -r path itself (the skip rule, src/genome_compare.c:138-141),
with a progress file (-p).

`write_inputs(dir)` builds the inputs deterministically (numpy PCG64, fixed seed) -- the tests call it, nothing but the
facts is committed.  Run as a script in the build container it also runs oracle/_ref/kmer_scrub_count (the reference,
built by oracle/Makefile from /root/reference/src) on them and writes tests/golden/cfg3_shape_facts.json:
md5 / line count of the TSV, the column sums, stderr, md5 of the progress file without its time stamps.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SEED = 0xC0F163
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")


def _dna(rng, n):
    return _ACGT[rng.integers(0, 4, n)]


def _mutate(rng, seq, rate):
    out = seq.copy()
    m = rng.random(out.size) < rate
    out[m] = _ACGT[rng.integers(0, 4, int(m.sum()))]
    return out


def _fasta(name, seq, width):
    b = seq.tobytes()
    return b">" + name + b"\n" + b"\n".join(b[i:i + width] for i in range(0, len(b), width)) + b"\n"


def write_inputs(d):
    """all inputs under directory d (relative paths inside the lists, as in the bundled example); returns the argv"""
    rng = np.random.default_rng(SEED)
    os.makedirs(os.path.join(d, "genomes"), exist_ok=True)
    os.makedirs(os.path.join(d, "reads"), exist_ok=True)
    contigs = [_dna(rng, 120_000), _dna(rng, 100_000), _dna(rng, 80_000)]
    for c in contigs:
        c[rng.choice(c.size, 3, replace=False)] = ord("N")
    with open(os.path.join(d, "strain.fa"), "wb") as f:
        for i, c in enumerate(contigs):
            f.write(_fasta(b"contig%d some description" % i, c, 60))
    whole = np.concatenate(contigs)
    # -A: 1000 genomes of 20 kbp; every 100th is a diverged copy of a strain segment, alternately reverse-complemented
    with open(os.path.join(d, "A.txt"), "w") as lst:
        for g in range(1000):
            if g % 100 == 7:
                a = int(rng.integers(0, whole.size - 20_000))
                seq = _mutate(rng, whole[a:a + 20_000], 0.01)
                if (g // 100) % 2:
                    seq = _COMP[seq][::-1]
            else:
                seq = _dna(rng, 20_000)
            p = "genomes/g%04d.fa" % g
            with open(os.path.join(d, p), "wb") as f:
                f.write(_fasta(b"g%d" % g, seq, 70 + g % 11))
            lst.write(p + "\n")
    # -B: 8 read files, FASTQ, every other one gzipped
    with open(os.path.join(d, "B.txt"), "w") as lst:
        for k in range(8):
            n = 20_000
            reads = _dna(rng, n * 150).reshape(n, 150)
            for i in np.flatnonzero(rng.random(n) < 0.02):
                a = int(rng.integers(0, whole.size - 150))
                r = _mutate(rng, whole[a:a + 150], 0.005)
                reads[i] = _COMP[r][::-1] if rng.random() < 0.5 else r
            flat = reads.reshape(-1)
            flat[rng.integers(0, flat.size, flat.size // 10_000)] = ord("N")
            body = b"".join(b"@read%d/%d\n%s\n+\n%s\n" % (k, i, reads[i].tobytes(), b"F" * 150) for i in range(n))
            p = "reads/mg%d.fq%s" % (k, ".gz" if k % 2 else "")
            if k % 2:
                with gzip.GzipFile(os.path.join(d, p), "wb", compresslevel=1, mtime=0) as f:
                    f.write(body)
            else:
                with open(os.path.join(d, p), "wb") as f:
                    f.write(body)
            lst.write(p + "\n")
    # -C: 5 genomes, the strain itself among them (listed with the -r path: skipped)
    with open(os.path.join(d, "C.txt"), "w") as lst:
        for k in range(5):
            if k == 2:
                lst.write("strain.fa\n")
                continue
            a = int(rng.integers(0, whole.size - 50_000))
            seq = np.concatenate([_mutate(rng, whole[a:a + 50_000], 0.02), _dna(rng, 30_000)])
            p = "genomes/drug%d.fa" % k
            with open(os.path.join(d, p), "wb") as f:
                f.write(_fasta(b"drug%d" % k, seq, 80))
            lst.write(p + "\n")
    return ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-C", "C.txt", "-p", "progress.txt"]


def progress_md5(path):
    """the progress file with the time stamps cut off (src/genome_compare.c:167-170 writes "<line>\\t<asctime>")"""
    lines = [l.split(b"\t")[0] for l in open(path, "rb").read().split(b"\n")]
    return hashlib.md5(b"\n".join(lines)).hexdigest()


def facts_of(stdout, stderr, d):
    cols = np.loadtxt([l for l in stdout.decode().split("\n")[1:] if l], dtype=np.int64, usecols=(1, 2, 3, 4), ndmin=2)
    return {"md5_stdout": hashlib.md5(stdout).hexdigest(), "lines": stdout.count(b"\n"),
            "column_sums": [int(x) for x in cols.sum(axis=0)], "stderr": stderr.decode(),
            "md5_progress_without_times": progress_md5(os.path.join(d, "progress.txt"))}


if __name__ == "__main__":
    import tempfile
    exe = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")
    assert os.access(exe, os.X_OK), "build oracle/_ref first (make -C oracle)"
    with tempfile.TemporaryDirectory(prefix="cfg3_shape_") as d:
        argv = write_inputs(d)
        p = subprocess.run([exe] + argv, cwd=d, capture_output=True)
        assert p.returncode == 0, p.stderr
        facts = {"producer": "oracle/_ref/kmer_scrub_count (unmodified reference)", "argv": argv, "returncode": p.returncode,
                 **facts_of(p.stdout, p.stderr, d)}
    out = os.path.join(REPO, "tests", "golden", "cfg3_shape_facts.json")
    json.dump(facts, open(out, "w"), indent=1)
    print(json.dumps(facts)[:400])
