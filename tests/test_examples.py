"""INTEGRATION.md section 2 as code: the reference-shaped main() over the C-ABI compiles against
include/strainer_kmer.h alone (CPU) and reproduces the reference's output (GPU)."""
import json
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "examples", "kmer_scrub_count_patched_main.c")


def _build(out):
    lib = os.path.join(REPO, "strainer2_amd", "lib")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(REPO, "include"), SRC, "-L" + lib, "-lstrainer_kmer",
                    "-Wl,-rpath," + lib, "-o", out], check=True)


def test_example_main_compiles_and_links(tmp_path):
    _build(str(tmp_path / "ksc_patched"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixed", "drug"])
def test_example_main_reproduces_reference_output(golden, tmp_path, name):
    exe = str(tmp_path / "ksc_patched")
    _build(exe)
    d = os.path.join(golden, "cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = [a if a not in ("progress.txt", "prog.txt") else str(tmp_path / "p") for a in meta["argv"]]
    p = subprocess.run([exe] + argv, cwd=d, capture_output=True)
    assert p.returncode == meta["returncode"]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
