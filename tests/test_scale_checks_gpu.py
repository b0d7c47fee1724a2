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
    p = subprocess.run([sys.executable, os.path.join(REPO, where, tool)], env=dict(os.environ, **env), capture_output=True, timeout=800)
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


def _shm_free_gb():
    import shutil
    try:
        return shutil.disk_usage("/dev/shm").free / 1e9
    except OSError:
        return 0.0


@pytest.mark.parametrize("pack", ["2", "0"])
def test_cfg3_at_spec_as_one_job_against_the_references_facts(pack):
    """(SK_LIST_PACK: every chunk uploaded packed by the decode threads / every chunk as bytes -- the default decides by the waits it sees)
    BASELINE configs[2] AT SPEC through bin/kmer_scrub_count as ONE job (the at-spec pins belong where the
    driver runs them): the real 1000 x 5 Mbp -A list (ten strain copies at 1 % divergence), the 67 x 1 M-read FASTQ -B files listed
    once, -C with the -r path among its five genomes, -p -- all four columns (sum, non-zero rows, max, md5 of the u32 vector in the
    reference's row order), stderr and the progress file against tests/golden/cfg3_full_facts.json, which the UNMODIFIED
    reference produced (tests/golden/make_cfg3_full_facts.py; the facts hold the one-pass metagenome column, so x 1 is as
    pinned as the x 10 of tools/cfg3_full.py's default: src/kmer_scrub_count.c:89-98).  26 GB of inputs under /dev/shm."""
    if _shm_free_gb() < 45:
        pytest.skip("needs 45 GB free under /dev/shm for the inputs and the table (%.0f GB free)" % _shm_free_gb())
    import json
    out = _run("cfg3_full.py", LIST_REPEAT="1", WORK="/dev/shm/sk_cfg3_test", SK_LIST_PACK=pack)
    rep = json.loads(out[out.index("{"):])
    assert rep["identical_to_the_reference"] is True and rep["differences"] == [], rep["differences"]
    assert rep["list_repeat"] == 1 and rep["bases_scanned_total"] == 5_000_000_000 + 10_050_000_000 + 20_000_000
    assert rep["got"]["columns"]["metagenome_count"]["md5_u32_le"] == "d0f6143d4bc3e13df9da93e2adb10193"


def test_cfg5_one_gpus_share_32_resident_strains_against_the_references_files():
    """One GPU's share of BASELINE configs[4]: `strain_detect -S` with all 32 strains of 5 Mbp resident (one union table) on the 1 Gbase
    prefix of the metagenome; the -o files of the two pinned strains must be, decompressed, byte for byte what the UNMODIFIED
    reference wrote for them (md5, lines, bytes, stdout, stderr: tests/golden/cfg5_share_facts.json, made by
    tests/golden/make_cfg5_share_facts.py; src/strain_detect.c:263-384)."""
    if _shm_free_gb() < 8:
        pytest.skip("needs 8 GB free under /dev/shm (%.0f GB free)" % _shm_free_gb())
    import json
    out = _run("sd_cfg5_share.py", ONLY_PREFIX="1", WORK="/dev/shm/sk_cfg5_test")
    rep = json.loads(out[out.index("{"):])
    assert rep["ok"] is True
    pin = [v for k, v in rep.items() if k.startswith("prefix_")][0]
    assert pin["returncode"] == 0 and len(pin["strains"]) == 2
    assert all(s["identical_to_the_reference"] for s in pin["strains"].values()), pin["strains"]
