"""The library's own gzip/DEFLATE decoder (strainer2_amd/csrc/sk_gzfast.h) against zlib's gzread on the same
files: every compression level, stored and fixed-Huffman blocks, long matches, repetitive and incompressible
data, multi-member files, all optional header fields, trailing garbage, truncated and corrupted streams, empty
members, and every .gz fixture of the repository.  The comparison program (tests/native/gzfast_check.c) is
built under AddressSanitizer + UBSan."""
import gzip
import os
import random
import subprocess
import zlib

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gz") / "gzfast_check")
    subprocess.run(["gcc"] + SAN + [os.path.join(REPO, "tests", "native", "gzfast_check.c"), "-lz", "-o", exe], check=True)
    return exe


def run(checker, files, pipe=False, par=0, seg=0):
    env = dict(ENV, SK_GZ_SEG=str(seg)) if seg else ENV
    p = subprocess.run([checker] + (["--pipe"] if pipe else []) + (["--par", str(par)] if par else []) + [str(f) for f in files],
                       env=env, capture_output=True)
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    assert b"MISMATCH" not in p.stdout
    return p.stdout.decode()


def payloads():
    rng = random.Random(12345)
    dna = "".join(rng.choice("ACGT") for _ in range(300_000))
    fastq = "".join(f"@r{i}\n{dna[i * 150:(i + 1) * 150]}\n+\n{''.join(rng.choice('ABCDEFGHIJ') for _ in range(150))}\n" for i in range(1500))
    return {
        "empty": b"",
        "one": b"x",
        "dna": dna.encode(),
        "fastq": fastq.encode(),
        "zeros": bytes(400_000),                                   # distance 1, length 258 runs
        "period3": b"abc" * 150_000,                               # short distances
        "random": bytes(rng.getrandbits(8) for _ in range(200_000)),   # incompressible: stored blocks at level 1+
        "text": (b"the quick brown fox jumps over the lazy dog\n" * 20_000),
        "far": (bytes(rng.getrandbits(8) for _ in range(32_000)) + b"Z") * 6,   # matches at the 32 KiB limit
        # symbol frequencies falling off geometrically: Huffman codes up to the 15-bit limit (subtables at full depth)
        "skewed": bytes(min(255, int(rng.expovariate(0.35))) for _ in range(400_000)),
        "skewed2": bytes((7 * min(36, int(rng.expovariate(0.5)))) & 255 for _ in range(300_000)),
    }


def test_levels_strategies_and_block_types(checker, tmp_path):
    files = []
    for name, data in payloads().items():
        for level in (0, 1, 2, 4, 6, 9):
            f = tmp_path / f"{name}.l{level}.gz"
            with gzip.GzipFile(f, "wb", compresslevel=level, mtime=0) as g:
                g.write(data)
            files.append(f)
        for strat, sname in ((zlib.Z_FIXED, "fixed"), (zlib.Z_HUFFMAN_ONLY, "huff"), (zlib.Z_RLE, "rle"), (zlib.Z_FILTERED, "filt")):
            c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
            f = tmp_path / f"{name}.{sname}.gz"
            f.write_bytes(c.compress(data) + c.flush())
            files.append(f)
        # many small blocks: a sync flush every few hundred bytes
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        out = b"".join(c.compress(data[i:i + 777]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 777)) + c.flush()
        f = tmp_path / f"{name}.sync.gz"
        f.write_bytes(out)
        files.append(f)
    out = run(checker, files)
    assert out.count(" OK ") == len(files)


def test_members_headers_and_garbage(checker, tmp_path):
    p = payloads()
    a, b = gzip.compress(p["fastq"], 6, mtime=0), gzip.compress(p["dna"], 1, mtime=0)
    (tmp_path / "two.gz").write_bytes(a + b)
    (tmp_path / "three_with_empty.gz").write_bytes(a + gzip.compress(b"", mtime=0) + b)
    (tmp_path / "garbage_after.gz").write_bytes(a + b"this is not gzip" * 10)
    (tmp_path / "zeros_after.gz").write_bytes(a + bytes(100))
    # FEXTRA + FNAME + FCOMMENT + FHCRC
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(p["text"]) + raw.flush()
    hdr = bytes([0x1F, 0x8B, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 3]) + (5).to_bytes(2, "little") + b"extra" + b"name.txt\0" + b"a comment\0"
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    (tmp_path / "all_fields.gz").write_bytes(hdr + body + zlib.crc32(p["text"]).to_bytes(4, "little") + (len(p["text"]) & 0xFFFFFFFF).to_bytes(4, "little"))
    (tmp_path / "plain.txt").write_bytes(p["fastq"])
    out = run(checker, sorted(tmp_path.iterdir()))
    assert out.count(" OK ") == 5 and "not gzip" in out


def test_truncated_and_corrupted(checker, tmp_path):
    p = payloads()
    good = gzip.compress(p["fastq"], 6, mtime=0)
    rng = random.Random(7)
    files = []
    for cut in (len(good) - 1, len(good) - 4, len(good) - 9, len(good) // 2, 100, 30, 19):
        f = tmp_path / f"cut{cut}.gz"
        f.write_bytes(good[:cut])
        files.append(f)
    for i in range(12):
        bad = bytearray(good)
        pos = rng.randrange(20, len(bad) - 8)
        bad[pos] ^= 1 << rng.randrange(8)
        f = tmp_path / f"flip{i}.gz"
        f.write_bytes(bytes(bad))
        files.append(f)
    crc = bytearray(good)
    crc[-6] ^= 0xFF
    (tmp_path / "bad_crc.gz").write_bytes(bytes(crc))
    files.append(tmp_path / "bad_crc.gz")
    out = run(checker, files)
    assert out.count("damaged") >= len(files) - 12          # (a flipped bit may happen to decode cleanly up to the CRC)


def test_every_gz_fixture_of_the_repository(checker):
    files = []
    for root, _d, names in os.walk(os.path.join(REPO, "tests", "golden")):
        files += [os.path.join(root, n) for n in names if n.endswith(".gz")]
    assert len(files) > 50
    out = run(checker, sorted(files))
    assert out.count(" OK ") == len(files)


def test_random_streams_all_encoder_settings(checker, tmp_path):
    """80 random payloads (random bytes, tiny alphabets, periodic, LZ-style self-copies at every distance, skewed
    alphabets, FASTQ) through random zlib settings: level 0-9, window 2^9-2^15, memLevel 1-9, every strategy,
    sync/full flushes in the middle, inputs fed in pieces"""
    rng = random.Random(2024)

    def gen(n):
        kind = rng.randrange(6)
        if kind == 0:
            return bytes(rng.getrandbits(8) for _ in range(n))
        if kind == 1:
            a = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
            return bytes(rng.choice(a) for _ in range(n))
        if kind == 2:
            seg = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 300)))
            return (seg * (n // len(seg) + 1))[:n]
        if kind == 3:
            out = bytearray()
            while len(out) < n:
                if out and rng.random() < 0.6:
                    d, ln = rng.randrange(1, min(len(out), 32768) + 1), rng.randrange(3, 300)
                    for _ in range(ln):
                        out.append(out[-d])
                else:
                    out += bytes(rng.choice(b"ACGTN\n@+IJK") for _ in range(rng.randrange(1, 50)))
            return bytes(out[:n])
        if kind == 4:
            lam = rng.uniform(0.05, 1.0)
            return bytes(min(255, int(rng.expovariate(lam))) for _ in range(n))
        return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(100)),
                                              bytes(rng.choice(b"FGHIJ#") for _ in range(100))) for i in range(n // 210 + 1))[:n]

    files = []
    for i in range(80):
        data = gen(rng.choice([0, 1, 2, 10, 100, 1000, 5000, 70000, 200000]))
        c = zlib.compressobj(rng.randrange(0, 10), zlib.DEFLATED, 16 + rng.randrange(9, 16), rng.randrange(1, 10), rng.choice([0, 1, 2, 3, 4]))
        blob, pos = b"", 0
        while pos < len(data):
            step = rng.choice([len(data), 1000, 37, 100000])
            blob += c.compress(data[pos:pos + step])
            pos += step
            if rng.random() < 0.2:
                blob += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        f = tmp_path / f"f{i}.gz"
        f.write_bytes(blob + c.flush())
        files.append(f)
    assert run(checker, files).count(" OK ") == len(files)
    assert run(checker, files, pipe=True).count(" OK ") == len(files)


@pytest.fixture(scope="module")
def checker_tsan(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gzt") / "gzfast_check_tsan")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", os.path.join(REPO, "tests", "native", "gzfast_check.c"),
                    "-lz", "-lpthread", "-o", exe], check=True)
    return exe


def test_helper_thread_pipe(checker, checker_tsan, tmp_path):
    """the same files pulled through sk_gzpipe.h (inflate on a helper thread), complete and abandoned half way,
    under ASan/UBSan and under ThreadSanitizer"""
    p = payloads()
    files = []
    for name in ("fastq", "zeros", "random", "empty", "one"):
        f = tmp_path / f"{name}.gz"
        f.write_bytes(gzip.compress(p[name], 6, mtime=0))
        files.append(f)
    (tmp_path / "two.gz").write_bytes(gzip.compress(p["fastq"], 1, mtime=0) + gzip.compress(p["dna"], 9, mtime=0))
    (tmp_path / "cut.gz").write_bytes(gzip.compress(p["fastq"], 6, mtime=0)[:50_000])
    files += [tmp_path / "two.gz", tmp_path / "cut.gz"]
    out = run(checker, files, pipe=True)
    assert out.count(" OK ") == 6 and out.count("damaged") == 1
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    q = subprocess.run([checker_tsan, "--pipe"] + [str(f) for f in files], env=env, capture_output=True)
    assert b"ThreadSanitizer" not in q.stderr and q.returncode == 0, q.stderr.decode()[-2000:]


def test_one_member_inflated_by_several_threads(checker, checker_tsan, tmp_path):
    """sk_gzpar.h: segments found by block search, decoded speculatively into symbols, stitched in order.  With
    segments of 64 bytes to 30 KB every route is taken many times (guess accepted, gap decoded first, segment
    decoded again, no block in a segment, member end inside a segment, second member, damage) and the bytes
    must still be zlib's, under ASan/UBSan and under ThreadSanitizer."""
    p = payloads()
    rng = random.Random(99)
    files = []
    for name in ("fastq", "dna", "text", "far", "skewed", "random", "zeros", "period3"):
        for level in (1, 6, 9):
            f = tmp_path / f"{name}.l{level}.gz"
            f.write_bytes(gzip.compress(p[name], level, mtime=0))
            files.append(f)
    for strat, sname in ((zlib.Z_FIXED, "fixed"), (zlib.Z_HUFFMAN_ONLY, "huff"), (zlib.Z_RLE, "rle")):
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
        f = tmp_path / f"fastq.{sname}.gz"
        f.write_bytes(c.compress(p["fastq"]) + c.flush())
        files.append(f)
    # pigz-style: an empty stored block (sync flush) every so often -- the boundary block the search does not see
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    f = tmp_path / "fastq.sync.gz"
    f.write_bytes(b"".join(c.compress(p["fastq"][i:i + 5000]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(p["fastq"]), 5000)) + c.flush())
    files.append(f)
    # fixed-Huffman and dynamic blocks taking turns (raw pieces of separate compressors, each ended by a full flush):
    # half the segment boundaries are followed by blocks the search skips, then by one it finds
    body, data = b"", p["fastq"][:300_000]
    for j, i in enumerate(range(0, len(data), 2500)):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED if j % 2 else zlib.Z_DEFAULT_STRATEGY)
        body += c.compress(data[i:i + 2500]) + c.flush(zlib.Z_FULL_FLUSH)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body += c.compress(b"the end\n") + c.flush()
    whole = data + b"the end\n"
    f = tmp_path / "fixed_dynamic_turns.gz"
    f.write_bytes(bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 3]) + body + zlib.crc32(whole).to_bytes(4, "little") + len(whole).to_bytes(4, "little"))
    files.append(f)
    a, b = gzip.compress(p["fastq"], 6, mtime=0), gzip.compress(p["dna"], 1, mtime=0)
    (tmp_path / "two.gz").write_bytes(a + b)
    (tmp_path / "garbage_after.gz").write_bytes(a + b"this is not gzip" * 1000)
    (tmp_path / "many_small.gz").write_bytes(b"".join(gzip.compress(p["fastq"][i:i + 20000], 6, mtime=0) for i in range(0, len(p["fastq"]), 20000)))
    files += [tmp_path / "two.gz", tmp_path / "garbage_after.gz", tmp_path / "many_small.gz"]
    good = len(files)
    for cut in (len(a) - 1, len(a) - 9, len(a) // 2, len(a) // 7):
        f = tmp_path / f"cut{cut}.gz"
        f.write_bytes(a[:cut])
        files.append(f)
    for i in range(10):
        bad = bytearray(a)
        bad[rng.randrange(20, len(bad) - 8)] ^= 1 << rng.randrange(8)
        f = tmp_path / f"flip{i}.gz"
        f.write_bytes(bytes(bad))
        files.append(f)
    total = [0, 0, 0]
    for seg in (64, 1500, 30000):
        out = run(checker, files, par=3, seg=seg)
        assert out.count(" OK ") >= good and out.count("damaged") >= 4
        stats = [int(x) for x in out.split("par: direct ")[1].replace("gap", "").replace("again", "").split()]
        assert stats[0] > 20 and stats[2] > 0, (seg, stats)
        total = [a + b for a, b in zip(total, stats)]
    assert min(total) > 0, total                                  # (direct, after a gap, decoded again)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", SK_GZ_SEG="1500")
    t = subprocess.run([checker_tsan, "--par", "4"] + [str(f) for f in files], env=env, capture_output=True)
    assert t.returncode == 0 and b"ThreadSanitizer" not in t.stderr and b"MISMATCH" not in t.stdout, (t.stdout[-1500:], t.stderr[-3000:])


def test_files_cut_short_give_exactly_zlibs_bytes(checker, tmp_path):
    """the reference parses whatever gzread can still decode of a truncated file, so the decoder must stop at the
    same symbol: 400 cut points over dynamic, fixed and stored blocks, block headers and the trailer (the checker
    insists on equal byte counts when zlib reports an early end of input)"""
    p = payloads()
    rng = random.Random(4321)
    blobs = {
        "fq6": gzip.compress(p["fastq"][:120_000], 6, mtime=0),
        "fq1": gzip.compress(p["fastq"][:120_000], 1, mtime=0),
        "stored": gzip.compress(p["random"][:70_000], 6, mtime=0),
        "text9": gzip.compress(p["text"][:200_000], 9, mtime=0),
    }
    c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    blobs["fixed"] = c.compress(p["fastq"][:60_000]) + c.flush()
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    blobs["sync"] = b"".join(c.compress(p["fastq"][i:i + 900]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, 60_000, 900)) + c.flush()
    files = []
    for name, blob in blobs.items():
        cuts = {len(blob) - k for k in range(1, 40)} | {rng.randrange(19, len(blob)) for _ in range(40)}
        for cut in sorted(cuts):
            f = tmp_path / f"{name}.{cut}.gz"
            f.write_bytes(blob[:cut])
            files.append(f)
    out = run(checker, files)
    assert out.count("damaged") == len(files) and out.count(": OK") == len(files)
    out = run(checker, files, par=3, seg=700)
    assert out.count("damaged") == len(files) and out.count(": OK") == len(files)
