"""How a -A/-B list is shared out: items dealt to ranks by size, big plain-text files cut into byte-range
pieces at CHECKED record boundaries (strainer2_amd/csrc/sk_host.c).  The identity that makes both harmless --
the ranks' count vectors add up to the unsharded one (src/genome_compare.c:220-223: counters only ever get +1) -- is
run on the CPU with the plain-C device double, world 2/4/8, several decode threads, under ASan+UBSan."""
import gzip
import os
import random
import subprocess

import pytest

import _synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(REPO, "tests", "native", f) for f in ("shard_check.c", "device_double.c")] + \
      [os.path.join(REPO, "strainer2_amd", "csrc", f) for f in ("sk_host.c", "sk_host_sd.c", "sk_host_cov.c")]


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shard") / "shard_check")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-DDOUBLE_NO_MAIN"] + SRC +
                   ["-lz", "-lpthread", "-o", out], check=True)
    return out


def _fastq(rng, strain, n, wrap_qual_at=False):
    out = []
    for i in range(n):
        L = rng.choice([31, 60, 150, 150, 150, 250])
        if rng.random() < 0.5:
            a = rng.randrange(0, len(strain) - L)
            s = strain[a:a + L]
        else:
            s = _synth.rand_dna(rng, L)
        q = bytes(rng.choice(b"@+>IIIIFF#") for _ in range(L)) if wrap_qual_at else b"I" * L   # quality lines that begin with @, + or >
        out.append(b"@r%d some comment\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


def _fasta(rng, strain, n, width):
    out = []
    for i in range(n):
        L = rng.randrange(100, 30000)
        a = rng.randrange(0, max(1, len(strain) - L))
        s = strain[a:a + L] if rng.random() < 0.6 else _synth.rand_dna(rng, L)
        out.append(b">c%d\n" % i + (b"\n".join(s[j:j + width] for j in range(0, len(s), width)) if width else s) + b"\n")
    return b"".join(out)


@pytest.fixture(scope="module")
def world_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard_data")
    rng = random.Random(2024)
    strain = _synth.rand_dna(rng, 60000)
    open(d / "strain.fa", "wb").write(b">s\n" + strain + b"\n")
    files = {
        "big.fq": _fastq(rng, strain, 6000),                       # by far the biggest: has to be cut or one rank does it all
        "tricky.fq": _fastq(rng, strain, 2500, wrap_qual_at=True),  # quality lines beginning with @ / + / >
        "wrapped.fa": _fasta(rng, strain, 60, 60),
        "oneline.fa": _fasta(rng, strain, 40, 0),                   # records longer than a piece
        "small1.fq": _fastq(rng, strain, 50),
        "small2.fa": _fasta(rng, strain, 3, 70),
        "crlf.fq": _fastq(rng, strain, 300).replace(b"\n", b"\r\n"),
    }
    for name, data in files.items():
        open(d / name, "wb").write(data)
    with gzip.open(d / "zipped.fq.gz", "wb") as f:
        f.write(_fastq(rng, strain, 800))
    names = list(files) + ["zipped.fq.gz", "small1.fq"]             # (a duplicate line: scanned twice, as the reference does)
    open(d / "list.txt", "w").write("".join(str(d / n) + "\n" for n in names))
    return d


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("split", ["20000", "150000", None])
def test_sharded_sums_equal_the_unsharded_scan(exe, world_files, world, split):
    env = dict(os.environ, SK_THREADS="3", ASAN_OPTIONS="detect_leaks=0")
    if split:
        env["SK_SPLIT_BYTES"] = split          # piece size: far below the 32 MiB floor, so that these small files ARE cut
    else:
        env["SK_NO_SPLIT"] = "1"               # dealing by size only
    p = subprocess.run([exe, str(world_files / "strain.fa"), str(world_files / "list.txt"), str(world)], env=env,
                       capture_output=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr[-2000:])
    tag, total, bases = p.stdout.split()
    assert tag == b"OK" and int(total) > 100000 and int(bases) > 1000000


@pytest.mark.parametrize("world", [2, 8])
def test_the_plan_does_not_follow_a_ranks_thread_count(exe, world_files, world):
    """the piece size came from the LOCAL thread count (SK_THREADS, cgroup quota ...), so ranks with
    different settings cut files differently and byte ranges were scanned twice or never.  The automatic piece size (no
    SK_SPLIT_BYTES; its 32 MiB floor lowered so that these small files are cut) is now a function of the list, the sizes
    and the world size: every rank scans with a different SK_THREADS and the sums still equal the unsharded scan."""
    env = dict(os.environ, SK_THREADS="3", ASAN_OPTIONS="detect_leaks=0", SK_SPLIT_FLOOR_BYTES="3000", SHARD_VARY_THREADS="1")
    p = subprocess.run([exe, str(world_files / "strain.fa"), str(world_files / "list.txt"), str(world)], env=env,
                       capture_output=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr[-2000:])
    assert p.stdout.startswith(b"OK")


def test_plan_hash_and_owners(exe, world_files):
    """skh_list_plan_hash / skh_list_plan_owners: the same for every thread count, different when a setting that does change
    the plan differs (what the ranks compare before scanning: SK_E_PLAN), every line owned, the big file shared"""
    def plan(**kw):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", SK_SPLIT_FLOOR_BYTES="3000", **kw)
        p = subprocess.run([exe, "--plan", str(world_files / "list.txt"), "4"], env=env, capture_output=True, timeout=60)
        assert p.returncode == 0, p.stderr
        f = p.stdout.split()
        return f[0], [int(x) for x in f[1:]]
    h1, o1 = plan(SK_THREADS="1")
    h16, o16 = plan(SK_THREADS="16")
    assert h1 == h16 and o1 == o16
    assert len(o1) == 9 and all(o in (0, 1, 2, 3, -2) for o in o1) and o1[0] == -2          # big.fq: pieces on several ranks
    h_other, _ = plan(SK_THREADS="1", SK_SPLIT_BYTES="20000")
    h_nosplit, o_ns = plan(SK_THREADS="1", SK_NO_SPLIT="1")
    assert len({h1, h_other, h_nosplit}) == 3 and -2 not in o_ns and set(o_ns) == {0, 1, 2, 3}


def test_a_file_that_cannot_be_cut_safely_fails_the_run(exe, tmp_path):
    """wrapped FASTQ whose quality lines imitate a header two lines before a '+' line: the guess lands inside a record,
    the check after the piece before it notices (parser not between two records) and the scan FAILS (SK_E_SPLIT = -9)
    instead of counting a different set of records; with SK_NO_SPLIT=1 the same list goes through"""
    rng = random.Random(5)
    strain = _synth.rand_dna(rng, 20000)
    open(tmp_path / "strain.fa", "wb").write(b">s\n" + strain + b"\n")
    recs = []
    for i in range(400):
        a = rng.randrange(0, len(strain) - 150)
        s = strain[a:a + 150]
        q = b"@" + b"I" * 49 + b"\n" + b"I" * 50 + b"\n" + b"+" + b"I" * 49          # three quality lines: "@..", "..", "+.."
        recs.append(b"@r%d\n%s\n%s\n%s\n+\n%s\n" % (i, s[:50], s[50:100], s[100:], q))
    open(tmp_path / "wrapped.fq", "wb").write(b"".join(recs))
    open(tmp_path / "list.txt", "w").write(str(tmp_path / "wrapped.fq") + "\n")
    env = dict(os.environ, SK_THREADS="3", ASAN_OPTIONS="detect_leaks=0", SK_SPLIT_BYTES="5000")
    p = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "2"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 1 and b"failed: -9" in p.stdout and b"could not be cut at record boundaries" in p.stderr
    env.pop("SK_SPLIT_BYTES")
    env["SK_NO_SPLIT"] = "1"
    p = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "2"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith(b"OK")


def test_a_record_that_ends_the_file_for_the_reference_is_not_read_past(exe, tmp_path):
    """a FASTQ record in the middle of a big file whose quality is longer than its sequence: the reference's reader returns -2
    there and the rest of the file is never read (src/kseq.h:205-209, src/genome_compare.c:203).  Cut into pieces, the later
    pieces would be counted all the same -- so the scan fails (SK_E_SPLIT) instead; uncut it stops where the reference
    stops: the same counts as the file's part before that record."""
    rng = random.Random(11)
    strain = _synth.rand_dna(rng, 20000)
    open(tmp_path / "strain.fa", "wb").write(b">s\n" + strain + b"\n")
    head = _fastq(rng, strain, 300)
    bad = b"@bad\n" + strain[100:250] + b"\n+\n" + b"I" * 170 + b"\n"
    tail = _fastq(rng, strain, 300)
    open(tmp_path / "whole.fq", "wb").write(head + bad + tail)
    open(tmp_path / "head.fq", "wb").write(head)
    open(tmp_path / "list.txt", "w").write(str(tmp_path / "whole.fq") + "\n")
    open(tmp_path / "list_head.txt", "w").write(str(tmp_path / "head.fq") + "\n")
    env = dict(os.environ, SK_THREADS="3", ASAN_OPTIONS="detect_leaks=0", SK_SPLIT_BYTES="8000")
    p = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "2"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 1 and b"failed: -9" in p.stdout and b"could not be cut at record boundaries" in p.stderr
    env.pop("SK_SPLIT_BYTES")
    env["SK_NO_SPLIT"] = "1"
    p = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "2"], env=env, capture_output=True, timeout=300)
    q = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list_head.txt"), "2"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0 and q.returncode == 0
    assert p.stdout.split()[:2] == q.stdout.split()[:2] and p.stdout.startswith(b"OK")


def test_one_process_scans_again_uncut_when_a_cut_does_not_hold(exe, tmp_path):
    """with ONE process a cut that fails its check must not fail the run -- the reference accepts these files.  The
    column and the progress file are put back and the list is scanned again uncut (sk_host.c: split_guard): the adversarial
    wrapped FASTQ and the file with a record that ends it for the reference both give, with three decode threads and tiny
    pieces, exactly what SK_NO_SPLIT=1 gives, and nothing on stderr."""
    rng = random.Random(17)
    strain = _synth.rand_dna(rng, 20000)
    open(tmp_path / "strain.fa", "wb").write(b">s\n" + strain + b"\n")
    recs = []
    for i in range(400):
        a = rng.randrange(0, len(strain) - 150)
        s = strain[a:a + 150]
        q = b"@" + b"I" * 49 + b"\n" + b"I" * 50 + b"\n" + b"+" + b"I" * 49
        recs.append(b"@r%d\n%s\n%s\n%s\n+\n%s\n" % (i, s[:50], s[50:100], s[100:], q))
    open(tmp_path / "wrapped.fq", "wb").write(b"".join(recs))
    head = _fastq(rng, strain, 300)
    bad = b"@bad\n" + strain[100:250] + b"\n+\n" + b"I" * 170 + b"\n"
    open(tmp_path / "whole.fq", "wb").write(head + bad + _fastq(rng, strain, 300))
    open(tmp_path / "ok.fq", "wb").write(_fastq(rng, strain, 500))
    open(tmp_path / "list.txt", "w").write("".join(str(tmp_path / n) + "\n" for n in ("ok.fq", "wrapped.fq", "whole.fq", "ok.fq")))
    base = dict(os.environ, SK_THREADS="3", ASAN_OPTIONS="detect_leaks=0")
    cut = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "1"], env=dict(base, SK_SPLIT_BYTES="5000"),
                         capture_output=True, timeout=300)
    uncut = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "1"], env=dict(base, SK_NO_SPLIT="1"),
                           capture_output=True, timeout=300)
    assert cut.returncode == 0 and uncut.returncode == 0, (cut.stdout, cut.stderr[-1500:])
    assert cut.stdout == uncut.stdout and cut.stdout.startswith(b"OK") and cut.stderr == b""
    said = subprocess.run([exe, str(tmp_path / "strain.fa"), str(tmp_path / "list.txt"), "1"], env=dict(base, SK_SPLIT_BYTES="5000", SK_TIMING="1"),
                          capture_output=True, timeout=300)
    assert said.stdout == uncut.stdout and b"did not hold; the list is scanned again uncut" in said.stderr      # (the fall-back did run)
