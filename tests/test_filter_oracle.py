"""The CPU restatements of scripts/kmer_scrub_filter.py and scripts/coverage_depth.py (oracle/ksf_oracle.c)
against the fixtures the reference's own scripts produced (tests/golden/make_golden_filter.py)."""
import json
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
FCASES = os.path.join(REPO, "tests", "golden", "filter_cases")
CCASES = os.path.join(REPO, "tests", "golden", "cov_cases")
BUNDLED = os.path.join(REPO, "tests", "golden", "bundled")


def _bin(name):
    p = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(p):
        subprocess.run(["make", "-C", ORACLE_DIR, name], check=True, stdout=subprocess.DEVNULL)
    return p


def check_case(binary, root, name, prepare=None):
    d = os.path.join(root, name)
    with open(os.path.join(d, "case.json")) as f:
        meta = json.load(f)
    if prepare:
        prepare(d)
    p = subprocess.run([binary] + meta["argv"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(os.path.join(d, "expected.stdout"), "rb") as f:
        want = f.read()
    with open(os.path.join(d, "expected.stderr"), "rb") as f:
        want_err = f.read()
    assert p.returncode == meta["returncode"], (name, p.stderr[-300:])
    assert p.stdout == want, name
    if meta["stderr_exact"]:
        assert p.stderr == want_err, name
    elif meta["returncode"] != 0:
        # the script died with a traceback: same exception class on the last line
        assert p.stderr.split(b":")[0] == want_err.split(b":")[0], (name, p.stderr, want_err)


def link_bundled_hits(d):
    dst = os.path.join(d, "Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208.kmer_hits.gz")
    if not os.path.exists(dst):
        import gzip
        with open(os.path.join(BUNDLED, "step3_expected.hits"), "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
            g.write(f.read())


@pytest.mark.parametrize("name", sorted(os.listdir(FCASES)))
def test_filter_oracle_matches_reference_script(name):
    check_case(_bin("ksf_oracle"), FCASES, name)


@pytest.mark.parametrize("name", sorted(os.listdir(CCASES)))
def test_coverage_oracle_matches_reference_script(name):
    check_case(_bin("kcd_oracle"), CCASES, name, link_bundled_hits if name == "bundled_step4" else None)
