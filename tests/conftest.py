import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build what the suite needs if it is not there yet (cross-compiles without a GPU).
    if not os.path.exists(os.path.join(REPO, "strainer2_amd", "lib", "libstrainer_kmer.so")) or \
       not os.path.exists(os.path.join(REPO, "strainer2_amd", "bin", "kmer_scrub_count")):
        subprocess.run(["make", "-C", os.path.join(REPO, "strainer2_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)
    # the checker (oracle restatement; the reference binaries where /root/reference exists) is rebuilt by
    # dependency, not by existence: a stale binary must never stand in for the current sources
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def repo():
    return REPO


@pytest.fixture(scope="session")
def golden():
    return os.path.join(REPO, "tests", "golden")


def has_gpu():
    try:
        import strainer2_amd as s
        s.KmerContext(0).close()
        return True
    except Exception:
        return False
