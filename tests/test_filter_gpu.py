"""The scrub filter on the GPU (reference scripts/kmer_scrub_filter.py): the kmer_scrub_filter program
against the fixtures the reference script produced, the device primitives (sk_filter_*) against numpy,
the program against the oracle on large tables, and the fused scan -> filter route."""
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import strainer2_amd as sk
from strainer2_amd import native

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FCASES = os.path.join(REPO, "tests", "golden", "filter_cases")
ORACLE_DIR = os.path.join(REPO, "oracle")
HEADER = "#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n"


def _oracle_bin(name):
    p = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(p):
        subprocess.run(["make", "-C", ORACLE_DIR, name], check=True, stdout=subprocess.DEVNULL)
    return p


@pytest.fixture(scope="module")
def ctx():
    c = sk.KmerContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", sorted(os.listdir(FCASES)))
def test_filter_program_matches_reference_script(name):
    d = os.path.join(FCASES, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    p = subprocess.run([sk.cli_path("kmer_scrub_filter")] + meta["argv"], cwd=d, capture_output=True)
    assert p.returncode == meta["returncode"], p.stderr.decode()[-500:]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    if meta["stderr_exact"]:
        assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
    elif meta["returncode"]:
        assert p.stderr


def _counts(rng, n, zero_frac, hi):
    v = rng.integers(1, hi, n)
    small = rng.random(n) < 0.7
    v[small] = rng.integers(1, 4, small.sum())
    v[rng.random(n) < zero_frac] = 0
    neg = rng.random(n) < 0.001
    v[neg] = -rng.integers(1, 2 ** 31, neg.sum())
    return v.astype(np.int64)


@pytest.mark.parametrize("n,seed", [(1, 1), (255, 2), (4096, 3), (4097, 4), (1_000_003, 5), (6_700_000, 6)])
def test_device_primitives_vs_numpy(ctx, n, seed):
    rng = np.random.default_rng(seed)
    pan, meta = _counts(rng, n, 0.6, 3000), _counts(rng, n, 0.3, 200000)
    gone = (rng.random(n) < 0.1).astype(np.uint8)
    f = native.ScrubFilter(ctx)
    f.load(pan, meta, gone)
    ps, ms = int(pan[pan > 0].sum()), int(meta[meta > 0].sum())
    assert f.sums() == (ps, ms, int((pan > 0).sum()), int((meta > 0).sum()), int(gone.sum()))
    # histogram windows
    for which, col in ((0, pan), (1, meta)):
        for lo, nb in ((1, 65536), (3, 100), (1000, 5000)):
            h = f.hist(which, lo, nb)
            pos = col[(col > 0) & (col >= lo)]
            want = np.bincount(np.minimum(pos - lo, nb), minlength=nb + 1)
            assert np.array_equal(h, want.astype(np.uint64))
    # independent thresholds
    for tp, tm in ((0, 0), (2, 10), (10 ** 9, 3)):
        want = (gone != 0) | ((pan > 0) & (pan > tp)) | ((meta > 0) & (meta > tm))
        assert np.array_equal(f.above(tp, tm), want.astype(np.uint8))
    # joint selection == stable descending sort of the scores
    score = np.zeros(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        if ms:
            score = np.maximum(score, np.where(meta > 0, meta / float(ms), 0.0))
        if ps:
            score = np.maximum(score, np.where(pan > 0, pan / float(ps), 0.0))
    alive = np.flatnonzero(gone == 0)
    order = alive[np.argsort(-score[alive], kind="stable")]
    for n_scrub in sorted({0, 1, len(alive) // 3, len(alive) // 2, max(len(alive) - 1, 0), len(alive)}):
        want = gone.copy()
        want[order[:n_scrub]] = 1
        got = f.joint(ps, ms, n_scrub)
        assert np.array_equal(got, want), (n, n_scrub, int((got != want).sum()))
    f.close()


def _write_table(path, rng, n, drug=False, dup=0):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = acgt[rng.integers(0, 4, (n, 31))]
    pan, meta = _counts(rng, n, 0.55, 2000), _counts(rng, n, 0.35, 50000)
    dr = (rng.random(n) < 0.1).astype(np.int64)
    lines = [HEADER.encode()]
    rows = list(range(n)) + [int(x) for x in rng.integers(0, n, dup)]
    for r in rows:
        k = keys[r].tobytes()
        lines.append(k + b"\t1\t%d\t%d" % (pan[r], meta[r]) + (b"\t%d\n" % dr[r] if drug else b"\n"))
    with gzip.GzipFile(path, "wb", compresslevel=1, mtime=0) as g:
        g.write(b"".join(lines))


@pytest.mark.parametrize("argv,drug,dup", [
    (["-m", "0.04"], False, 0), (["-m", "0.3"], True, 0), (["-m", "0.25", "-i"], False, 50), (["-m", "0.11"], True, 200),
    (["-m", "0.4", "-i"], True, 0),
])
def test_filter_program_vs_oracle_large(tmp_path, argv, drug, dup):
    rng = np.random.default_rng(len(argv) * 7 + dup + int(drug))
    t = str(tmp_path / "t.gz")
    _write_table(t, rng, 300_000, drug, dup)
    want = subprocess.run([_oracle_bin("ksf_oracle"), "-s", t] + argv, capture_output=True)
    got = subprocess.run([sk.cli_path("kmer_scrub_filter"), "-s", t] + argv, capture_output=True)
    assert got.returncode == want.returncode == 0
    assert got.stdout == want.stdout
    assert got.stderr == want.stderr


def test_independent_walk_crosses_histogram_windows(tmp_path):
    """more than 65536 thresholds: the walk refills its histogram window"""
    rng = np.random.default_rng(99)
    n = 2000
    keys = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 31))]
    with gzip.GzipFile(tmp_path / "t.gz", "wb", mtime=0) as g:
        for r in range(n):
            g.write(keys[r].tobytes() + b"\t1\t%d\t%d\n" % (rng.integers(1, 5), rng.integers(60000, 200000)))
    argv = ["-s", str(tmp_path / "t.gz"), "-m", "0.5", "-i"]
    want = subprocess.run([_oracle_bin("ksf_oracle")] + argv, capture_output=True)
    got = subprocess.run([sk.cli_path("kmer_scrub_filter")] + argv, capture_output=True)
    assert want.returncode == 0 and want.stderr.count(b"\n") > 70000
    assert (got.returncode, got.stdout, got.stderr) == (0, want.stdout, want.stderr)


@pytest.mark.parametrize("case,extra", [("mixed", ["--scrub", "0.3"]), ("mixed", ["--scrub=0.1", "--independent"]),
                                        ("drug", ["--scrub", "0.0"]), ("iupac_strain", ["--scrub", "0.5"])])
def test_fused_scan_then_filter_equals_two_steps(golden, tmp_path, case, extra):
    """kmer_scrub_count --scrub m == kmer_scrub_count | gzip | kmer_scrub_filter.py -s - -m m (the reference's
    step 1 output is the golden TSV; the script's restatement is the oracle)."""
    d = os.path.join(golden, "cases", case)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = [a if a not in ("progress.txt", "prog.txt") else str(tmp_path / "p") for a in meta["argv"]]
    with gzip.GzipFile(tmp_path / "t.gz", "wb", mtime=0) as g:
        g.write(open(os.path.join(d, "expected.stdout"), "rb").read())
    m = extra[0].split("=")[1] if "=" in extra[0] else extra[1]
    want = subprocess.run([_oracle_bin("ksf_oracle"), "-s", str(tmp_path / "t.gz"), "-m", m] +
                          (["-i"] if "--independent" in extra else []), capture_output=True)
    got = subprocess.run([sk.cli_path()] + argv + extra, cwd=d, capture_output=True)
    assert got.returncode == want.returncode
    assert got.stdout == want.stdout
    assert got.stderr.endswith(want.stderr)


def test_bundled_steps_1_and_2(golden, tmp_path):
    """test/example.sh steps 1+2 on the bundled data: the separate programs and the fused route both give
    the scrubbed k-mer list the reference's own script produced (tests/golden/bundled/strains/B8.scrubbed_kmers.gz)."""
    b = os.path.join(golden, "bundled")
    facts = json.load(open(os.path.join(b, "step1_facts.json")))
    want = gzip.open(os.path.join(b, "strains", "B8.scrubbed_kmers.gz"), "rb").read()
    assert hashlib.md5(want).hexdigest() == "fe981fa571be70e602875ac3463ecdac"
    fused = subprocess.run([sk.cli_path()] + facts["argv"] + ["--scrub", "0.01"], cwd=b, capture_output=True)
    assert fused.returncode == 0 and fused.stderr == b""
    assert fused.stdout == want
    table = tmp_path / "counts.gz"
    p1 = subprocess.Popen([sk.cli_path()] + facts["argv"], cwd=b, stdout=subprocess.PIPE)
    with open(table, "wb") as f:
        p2 = subprocess.run(["gzip", "-1"], stdin=p1.stdout, stdout=f)
    assert p1.wait() == 0 and p2.returncode == 0
    two = subprocess.run([sk.cli_path("kmer_scrub_filter"), "-s", str(table), "-m", "0.01"], capture_output=True)
    assert two.returncode == 0 and two.stdout == want
