"""The by-hand scale checks of tools/ at a size that fits the suite: odd strain shapes scanned against themselves (every
row counted exactly as often as it occurs), and both programs against the oracle programs on data that keeps the
byte-string path busy (IUPAC letters, U, N, lower case; PE, SE, interleaved; a third of the reads shorter than k)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, **env):
    where = "tools" if os.path.exists(os.path.join(REPO, "tools", tool)) else os.path.join("tests", "checks")
    p = subprocess.run([sys.executable, os.path.join(REPO, where, tool)], env=dict(os.environ, **env), capture_output=True, timeout=600)
    assert p.returncode == 0, (p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    return p.stdout.decode()


def test_odd_strain_shapes_scanned_against_themselves():
    out = _run("self_scan_check.py", STRAIN_BP="300000")
    assert out.count("every row counted as often as it occurs: True") == 7


@pytest.mark.parametrize("env", [{"SEED": "5"}, {"SEED": "6", "SHORT": "1"}])
def test_programs_against_oracle_programs_with_a_busy_byte_string_path(env):
    out = _run("iupac_diff_check.py", **env)
    assert out.count("identical") == 4 and "DIFFERENT" not in out


def test_many_strains_in_one_pass_against_the_oracle_strain_by_strain():
    """`strain_detect -S` on random worlds of 2-6 related strains, SE/PE/PEI files (plain, .gz, FASTA, FASTQ), a third of the
    reads below k, tiny chunks, several parser threads, union table on and off -- every strain's outfile against the oracle
    program's (tests/checks/sd_multi_diff_check.py; a hunt over more seeds is run by hand with SEEDS=a..b)"""
    out = _run("sd_multi_diff_check.py", SEEDS="0..9")
    assert out.count("identical") == 11 and "differs" not in out
