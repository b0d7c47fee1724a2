#!/usr/bin/env python3
"""tests/golden/make_two_expansions_facts.py -- pin the row order of a key set that makes the reference's hash table grow TWICE.

The reference's table starts at 8,000,000 slots and doubles when the insert count passes half its size (src/BIO_hash.c:129-139,
39-61; src/genome_compare.h:20): the 4,000,001st and the 8,000,001st distinct k-mer each re-insert everything, and the order in
which BIO_getHashKeys lists the rows afterwards (src/BIO_hash.c:174-188) is what the host's replay must reproduce
(replay_slot_order, strainer2_amd/csrc/sk_host.c).  The bundled example (6.7 M keys) and configs[1..2] (5 M) grow once.
This runs the UNMODIFIED reference program (oracle/_ref/kmer_scrub_count) on a synthetic 9.2 Mbp strain with empty lists and
records facts about its table: rows, md5 of the k-mer column packed to 62 bits (A0 C1 G2 T3, first base highest) as
little-endian u64 in row order, md5 of the reference_count column as little-endian u32, md5 of the whole stdout.
Only data is committed (tests/golden/two_expansions_facts.json); build container only.

  python3 tests/golden/make_two_expansions_facts.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")
SEED, CONTIGS, CONTIG_BP, LINE = 0x2EA9, 92, 100_020, 60


def write_strain(path):
    """92 contigs of 100,020 bp, uniform ACGT, 60 columns, an N every ~1 Mbp and one lower-case contig (tests use the same function)"""
    rng = np.random.default_rng(SEED)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for c in range(CONTIGS):
            seq = acgt[rng.integers(0, 4, size=CONTIG_BP)].copy()
            if c % 10 == 3:
                seq[int(rng.integers(1000, CONTIG_BP - 1000))] = ord("N")
            if c == 7:
                seq = np.frombuffer(seq.tobytes().lower(), dtype=np.uint8)
            f.write(b">contig%d two expansions\n" % c)
            rows = seq.reshape(-1, LINE)
            out = np.empty((rows.shape[0], LINE + 1), dtype=np.uint8)
            out[:, :LINE] = rows
            out[:, LINE] = 10
            f.write(out.tobytes())


def main():
    work = tempfile.mkdtemp(prefix="sk_two_exp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    strain = os.path.join(work, "strain.fa")
    write_strain(strain)
    for n in ("A.txt", "B.txt"):
        open(os.path.join(work, n), "w").close()
    out = os.path.join(work, "out.tsv")
    with open(out, "wb") as f:
        p = subprocess.run([EXE, "-r", strain, "-A", os.path.join(work, "A.txt"), "-B", os.path.join(work, "B.txt")], stdout=f, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    buf = np.fromfile(out, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    starts = nl[:-1] + 1                                         # (rows: every line but the header)
    kmers = buf[starts[:, None] + np.arange(31)[None, :]]
    code = np.zeros(256, dtype=np.uint64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    assert np.isin(kmers, np.frombuffer(b"ACGT", dtype=np.uint8)).all(), "a key with another letter: pack them otherwise"
    packed = np.zeros(len(starts), dtype=np.uint64)
    for i in range(31):
        packed = (packed << np.uint64(2)) | code[kmers[:, i]]
    import pandas as pd
    ref = pd.read_csv(out, sep="\t", header=0, usecols=[1], dtype=np.int64, engine="c").iloc[:, 0].to_numpy().astype(np.uint32)
    assert len(ref) == len(packed)
    facts = {"producer": "oracle/_ref/kmer_scrub_count (the unmodified reference), -r strain with empty -A and -B lists",
             "strain": {"seed": SEED, "contigs": CONTIGS, "contig_bp": CONTIG_BP, "line": LINE, "md5": hashlib.md5(open(strain, "rb").read()).hexdigest()},
             "rows": int(len(packed)), "expansions": "at the 4,000,001st and the 8,000,001st distinct k-mer: 8 M -> 16 M -> 32 M slots",
             "packed_keys_md5_u64_le": hashlib.md5(packed.astype("<u8").tobytes()).hexdigest(),
             "reference_count_md5_u32_le": hashlib.md5(ref.astype("<u4").tobytes()).hexdigest(),
             "reference_count_sum": int(ref.astype(np.uint64).sum()),
             "stdout_md5": hashlib.md5(buf.tobytes()).hexdigest(), "stdout_bytes": int(buf.size)}
    json.dump(facts, open(os.path.join(REPO, "tests", "golden", "two_expansions_facts.json"), "w"), indent=1)
    print(json.dumps(facts, indent=1))
    import shutil
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
