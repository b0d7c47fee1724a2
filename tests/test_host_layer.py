"""Host-layer logic and the C-ABI surface; runs without a GPU (no compute calls)."""
import ctypes
import json
import os
import random
import re
import subprocess

import numpy as np
import pytest

import _oracle
import _synth
import strainer2_amd as sk
from strainer2_amd import native


def test_library_exports_every_declared_symbol(repo):
    hdr = open(os.path.join(repo, "include", "strainer_kmer.h")).read()
    declared = set(re.findall(r"\b(skh?_[a-z0-9_]+)\s*\(", hdr)) - {"skh_sink_fn"}
    assert declared == set(native.ABI_SYMBOLS), declared ^ set(native.ABI_SYMBOLS)
    lib = ctypes.CDLL(sk.library_path())
    for name in sorted(declared):
        assert getattr(lib, name) is not None


def test_no_oracle_in_product(repo):
    """The product tree must not reference the oracle (it is test infrastructure)."""
    for root, _dirs, files in os.walk(os.path.join(repo, "strainer2_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "kso_" not in text and "oracle" not in text.lower().replace("no oracle", ""), os.path.join(root, f)


@pytest.mark.parametrize("name", ["mixed", "drug", "iupac_strain", "truncated_fastq", "contig30"])
def test_keyset_rows_match_reference_order(golden, name):
    d = os.path.join(golden, "cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    r = meta["argv"][meta["argv"].index("-r") + 1]
    ks = sk.Keyset.from_file(os.path.join(d, r))
    exp = [ln.split(b"\t") for ln in open(os.path.join(d, "expected.stdout"), "rb").read().split(b"\n")[1:] if ln]
    assert ks.nrows == len(exp)
    assert ks.keys() == [e[0] for e in exp]
    assert ks.first_count().tolist() == [int(e[1]) for e in exp]


def test_keyset_bundled_strain_order_and_doubling(golden):
    """6.7 M keys: exercises the 8 M -> 16 M doubling replay (src/BIO_hash.c:39-61)."""
    b = os.path.join(golden, "bundled")
    facts = json.load(open(os.path.join(b, "step1_facts.json")))
    ks = sk.Keyset.from_file(os.path.join(b, facts["argv"][1]))
    assert ks.nrows == facts["stdout_lines"] - 1 == 6698540
    assert ks.final_slots == 16000000
    assert int(ks.first_count().sum()) == facts["column_sums"][0]
    keys = ks.keys()
    assert [k.decode() for k in keys[:5]] == [r.split("\t")[0] for r in facts["first_rows"][1:6]]
    assert [k.decode() for k in keys[-3:]] == [r.split("\t")[0] for r in facts["last_rows"]]
    # full order against the oracle's table
    t = _oracle.OracleTable()
    assert t.build_file(os.path.join(b, facts["argv"][1])) == 0
    okeys, ocounts = t.rows()
    assert okeys == keys
    assert np.array_equal(ocounts[:, 0], ks.first_count())


def test_keyset_short_record_policy(golden):
    d = os.path.join(golden, "cases", "short_contig")
    ks = sk.Keyset.from_file(os.path.join(d, "strain.fa"))
    assert ks.short_records == 1                       # the reference crashes here; we skip + warn
    t = _oracle.OracleTable()
    assert t.build_file(os.path.join(d, "strain.fa"), short_policy=0) == -2
    t2 = _oracle.OracleTable()
    assert t2.build_file(os.path.join(d, "strain.fa"), short_policy=1) == 0
    assert t2.rows()[0] == ks.keys()


@pytest.mark.parametrize("seed", [11, 12])
def test_keyset_fuzz_vs_oracle(seed):
    rng = random.Random(seed)
    strain = _synth.rand_dna(rng, 2000)
    stream = _synth.fuzz_stream(rng, strain, 60, p_junk=0.01, min_len=30, max_len=300)
    ks = sk.Keyset.from_stream(stream, initial_slots=50)        # small table: many doublings
    t = _oracle.OracleTable(capacity=50)
    assert t.build_stream(stream) == 0
    okeys, ocounts = t.rows()
    assert ks.final_slots == t.capacity
    assert ks.keys() == okeys
    assert np.array_equal(ks.first_count(), ocounts[:, 0])


def _files(golden):
    out = []
    for root, _d, files in os.walk(os.path.join(golden, "cases")):
        out += [os.path.join(root, f) for f in files if f.endswith((".fa", ".fasta", ".fq", ".gz", ".fx"))]
    return sorted(out)


def test_reader_matches_oracle_records(golden):
    """Record grammar (src/kseq.h:166-211): same records out of every fixture file."""
    for path in _files(golden):
        data, nrec, _st = _oracle.decode_file(path)
        want = [r for r in data.split(b"\n")[:-1] if len(r) >= 31]
        chunks, got_nrec, bases = sk.decode_file(path)
        got = b"".join(chunks).split(b"\n")[:-1]
        assert got == want, path
        assert got_nrec == nrec, path
        assert bases == len(data) - nrec, path


def test_reader_cuts_long_records_with_overlap(tmp_path):
    rng = random.Random(5)
    seq = _synth.rand_dna(rng, 20000)
    p = tmp_path / "long.fa"
    p.write_bytes(b">x\n" + seq + b"\n>y\n" + seq[:40] + b"\n")
    chunks, nrec, bases = sk.decode_file(str(p), chunk_bytes=4096)
    assert nrec == 2 and bases == 20040
    pieces = b"".join(chunks).split(b"\n")[:-1]
    # windows of all pieces == windows of the records, each exactly once
    def windows(rs):
        w = []
        for r in rs:
            w += [r[i:i + 31] for i in range(len(r) - 30)]
        return sorted(w)
    assert windows(pieces) == windows([seq, seq[:40]])


def test_cli_usage_and_open_errors_need_no_gpu(golden, tmp_path):
    exe = sk.cli_path()
    d = os.path.join(golden, "cases", "missing_flag")
    meta = json.load(open(os.path.join(d, "case.json")))
    p = subprocess.run([exe] + meta["argv"], cwd=d, capture_output=True)
    assert p.returncode == 1 and p.stdout == b""
    assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
    p = subprocess.run([exe, "-r", "nope.fa", "-A", "a", "-B", "b"], cwd=str(tmp_path), capture_output=True)
    assert p.returncode == 1
    assert p.stderr == b"could not read file nope.fa GEN_hash_sequences_set_count_vec()\n"
    p = subprocess.run([exe, "-r", "nope.fa", "-A", "a", "-B", "b", "-p", "/nonexistent/dir/p"], cwd=str(tmp_path), capture_output=True)
    assert p.returncode == 1 and p.stderr == b"could not open progress file /nonexistent/dir/p\n"


def test_context_fails_loudly_without_gpu():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(sk.SKError) as e:
        sk.KmerContext(0)
    assert e.value.code == native.SK_E_NODEVICE
