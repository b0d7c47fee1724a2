"""The programs on a .gz file big enough for the several-thread inflate at its real segment size (2 MiB):
kmer_scrub_count and strain_detect must write the same bytes whether the file is read as plain text, inflated by
the one helper thread, or by four threads (strainer2_amd/csrc/sk_gzpar.h); the plain-text run is the one the
other GPU tests tie to the oracle and the reference."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(REPO, "strainer2_amd", "bin")


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    d = tmp_path_factory.mktemp("gzin")
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    strain = acgt[rng.integers(0, 4, 60_000)]
    (d / "strain.fa").write_bytes(b">s\n" + strain.tobytes() + b"\n")
    n = 260_000
    reads = acgt[rng.integers(0, 4, (n, 150))]
    for i in np.flatnonzero(rng.random(n) < 0.05):
        a = int(rng.integers(0, len(strain) - 150))
        reads[i] = strain[a:a + 150]
    qual = np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8)[rng.integers(0, 11, (n, 150))]
    text = b"".join(b"@read%d\n%s\n+\n%s\n" % (j, reads[j].tobytes(), qual[j].tobytes()) for j in range(n))
    (d / "reads.fq").write_bytes(text)
    blob = gzip.compress(text, 1, mtime=0)
    assert len(blob) * 3 // 5 > 5 * (2 << 20)         # both files: more than four segments, the parallel route is taken
    (d / "reads.fq.gz").write_bytes(blob)
    (d / "cut.fq.gz").write_bytes(blob[:len(blob) * 3 // 5])      # a download that stopped
    (d / "cut.fq").write_bytes(_zlib_prefix(blob[:len(blob) * 3 // 5]))
    # informative k-mers for strain_detect: every 50th k-mer of the strain
    s = strain.tobytes()
    (d / "inf.txt").write_bytes(b"#informative\n" + b"\n".join(s[i:i + 31] for i in range(0, len(s) - 31, 50)) + b"\n")
    return d


def _zlib_prefix(blob):
    """what zlib can still decode of a gzip file cut short (what the reference gets to see)"""
    import zlib
    z = zlib.decompressobj(31)
    return z.decompress(blob)


def _count(d, name, env):
    lst = d / (name + ".list")
    lst.write_text(str(d / name) + "\n")
    p = subprocess.run([os.path.join(BIN, "kmer_scrub_count"), "-r", str(d / "strain.fa"), "-A", str(lst), "-B", str(lst)],
                       env=dict(os.environ, **env), capture_output=True, check=True)
    return hashlib.md5(p.stdout).hexdigest(), p.stderr


def _detect(d, name, env, out):
    subprocess.run([os.path.join(BIN, "strain_detect"), "-r", str(d / "strain.fa"), "-a", str(d / "inf.txt"), "-b", str(d / name),
                    "-t", "SE", "-o", str(out)], env=dict(os.environ, **env), capture_output=True, check=True)
    # (every hit line names the read file: take the name out before comparing)
    return hashlib.md5(gzip.open(out, "rb").read().replace(str(d / name).encode(), b"FILE")).hexdigest()


@pytest.mark.parametrize("plain, packed", [("reads.fq", "reads.fq.gz"), ("cut.fq", "cut.fq.gz")])
def test_same_table_from_text_one_thread_and_four(world, plain, packed):
    want = _count(world, plain, {"SK_THREADS": "4"})
    assert _count(world, packed, {"SK_THREADS": "4", "SK_GZ_THREADS": "1"}) == want
    assert _count(world, packed, {"SK_THREADS": "8", "SK_GZ_THREADS": "4"}) == want
    assert _count(world, packed, {"SK_THREADS": "8"}) == want                         # the default split of the budget
    assert _count(world, packed, {"SK_THREADS": "1"}) == want                         # the strict sequence, no helper thread


@pytest.mark.parametrize("plain, packed", [("reads.fq", "reads.fq.gz"), ("cut.fq", "cut.fq.gz")])
def test_same_hit_list_from_text_one_thread_and_four(world, tmp_path, plain, packed):
    want = _detect(world, plain, {}, tmp_path / "a.gz")
    assert _detect(world, packed, {"SK_GZ_THREADS": "1"}, tmp_path / "b.gz") == want
    assert _detect(world, packed, {"SK_GZ_THREADS": "4"}, tmp_path / "c.gz") == want
    assert _detect(world, packed, {}, tmp_path / "d.gz") == want


def _has_device_inflate():
    import ctypes
    try:
        ctypes.CDLL(os.path.join(REPO, "strainer2_amd", "lib", "libstrainer_kmer.so")).sk_inflate_gz
        return True
    except (AttributeError, OSError):
        return False


@pytest.mark.skipif(not _has_device_inflate(), reason="the device-side gzip decoder is only in an experiments build (make -C strainer2_amd/csrc EXPERIMENTS=1): "
                                                      "measured, it loses to the host's decode threads (DESIGN.md section 7)")
def test_same_table_with_the_inflate_on_the_device(world, tmp_path):
    """SK_GPU_INFLATE=1 (experiments build only, sk_inflate.hip): a .gz of one member and dynamic blocks is inflated on the device -- speculative
    segment starts, every guess and the chain of segments checked, CRC-32 checked -- and must give the table the text gives; anything
    the device path does not take (a file cut short, two members, stored blocks, a flipped bit, FASTA with long lines) goes to the
    host decoder as before, with the same result as without the switch."""
    import zlib
    d = world
    want = _count(d, "reads.fq", {"SK_THREADS": "4"})
    got = _count(d, "reads.fq.gz", {"SK_THREADS": "4", "SK_GPU_INFLATE": "2"})
    assert got[0] == want[0] and got[1].count(b"inflated on the device") == 2          # (the -A and the -B list name the file)
    text = (d / "reads.fq").read_bytes()
    text = text[:text.rfind(b"\n@read", 0, 24_000_000) + 1]        # (a third of it is plenty for the variants)
    (d / "part.fq").write_bytes(text)
    want_part = _count(d, "part.fq", {"SK_THREADS": "4"})
    cases = {}
    cases["two_members.fq.gz"] = gzip.compress(text[:len(text) // 2], 6, mtime=0) + gzip.compress(text[len(text) // 2:], 6, mtime=0)
    cases["stored.fq.gz"] = gzip.compress(text[:3_000_000], 0, mtime=0)
    cases["level9.fq.gz"] = gzip.compress(text, 9, mtime=0)
    blob = bytearray(gzip.compress(text, 4, mtime=0))
    blob[len(blob) // 2] ^= 0x10
    cases["flipped.fq.gz"] = bytes(blob)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    cases["flushed.fq.gz"] = co.compress(text[:5_000_000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(text[5_000_000:]) + co.flush()   # an empty stored block inside
    for name, data in cases.items():
        (d / name).write_bytes(data)
        a = _count(d, name, {"SK_THREADS": "4"})
        b = _count(d, name, {"SK_THREADS": "4", "SK_GPU_INFLATE": "2"})
        assert a[0] == b[0], name
        assert (b"inflated on the device" in b[1]) == (name == "level9.fq.gz") and b"left to the host decoder" in b[1] or name == "level9.fq.gz", (name, b[1][-300:])
    assert _count(d, "level9.fq.gz", {"SK_THREADS": "4", "SK_GPU_INFLATE": "1"})[0] == want_part[0]
    assert _count(d, "cut.fq.gz", {"SK_THREADS": "4", "SK_GPU_INFLATE": "1"}) == _count(d, "cut.fq", {"SK_THREADS": "4"})
