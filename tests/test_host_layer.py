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

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def test_keyset_order_after_two_expansions_is_the_references(golden, tmp_path):
    """9.2 M keys: the reference's table grows at its 4,000,001st AND its 8,000,001st distinct k-mer (src/BIO_hash.c:129-139,39-61),
    every earlier case grows once at most at the real initial size.  Row order (the keys packed to 62 bits, md5) and the
    reference_count column of the host's replay and of the oracle against the UNMODIFIED reference's table
    (tests/golden/two_expansions_facts.json)."""
    import hashlib
    path, facts = _synth.two_expansions_strain(golden, tmp_path)
    if path is None:
        pytest.skip("this numpy's generator draws another strain than the one the facts were made from")
    ks = sk.Keyset.from_file(path)
    assert ks.nrows == facts["rows"] and ks.final_slots == 32000000 and ks.nwide == 0
    assert hashlib.md5(ks.packed().astype("<u8").tobytes()).hexdigest() == facts["packed_keys_md5_u64_le"]
    assert hashlib.md5(ks.first_count().astype("<u4").tobytes()).hexdigest() == facts["reference_count_md5_u32_le"]
    # the oracle restatement at the same size
    t = _oracle.OracleTable()
    assert t.build_file(path) == 0 and t.size == facts["rows"] and t.capacity == 32000000
    okeys, ocounts = t.rows()
    code = np.zeros(256, dtype=np.uint64)
    code[list(b"ACGT")] = np.arange(4, dtype=np.uint64)
    kb = np.frombuffer(b"".join(okeys), dtype=np.uint8).reshape(-1, 31)
    packed = np.zeros(len(okeys), dtype=np.uint64)
    for i in range(31):
        packed = (packed << np.uint64(2)) | code[kb[:, i]]
    assert np.array_equal(packed, ks.packed())
    assert np.array_equal(ocounts[:, 0], ks.first_count())
    t.close()
    ks.close()


@pytest.mark.parametrize("simd", ["1", "0"])
def test_host_pre_pack_is_the_scan_kernels_decode(simd, monkeypatch):
    """sk_pack_stream (the host-side 2-bit pre-pack, sk_pack.h) against the definition it shares with the scan kernel's first phase
    (sk_decode16): per 16-byte chunk a code word -- A 0, C 1, G 2, T 3 in either case, first byte highest, 0 for any other byte --
    and a mask of the bytes that are no A/C/G/T; `odd` iff some byte is neither A/C/G/T, N/n nor a newline.  Every byte value, ragged
    ends, the vector path and the table path."""
    monkeypatch.setenv("SK_PACK_SIMD", simd)
    import subprocess
    import sys
    code = r"""
import sys, random
import numpy as np
sys.path.insert(0, %r)
import strainer2_amd as sk
rng = random.Random(5)
lut = {ord(c): i for i, c in enumerate("ACGT")}
lut.update({ord(c): i for i, c in enumerate("acgt")})
fine = set(b"ACGTacgtNn\n")
for rep in range(300):
    n = rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 1000, 4099])
    kind = rep %% 3
    b = bytes(rng.randrange(256) if kind == 0 else rng.choice(b"ACGTacgtNn\n") if kind == 1 else (rng.choice(b"ACGT") if rng.random() < 0.97 else rng.randrange(256)) for _ in range(n))
    packed, odd = sk.pack_stream(b)
    nch = (n + 15) // 16
    assert len(packed) == 6 * nch
    codes = np.frombuffer(packed[:4 * nch].tobytes(), dtype="<u4")
    inv = np.frombuffer(packed[4 * nch:].tobytes(), dtype="<u2")
    for g in range(nch):
        c = m = 0
        for i in range(16):
            at = 16 * g + i
            v = lut.get(b[at]) if at < n else None
            c = (c << 2) | (v or 0)
            if v is None:
                m |= 1 << i
        assert (int(codes[g]), int(inv[g])) == (c, m), (rep, g)
    assert odd == any(x not in fine for x in b), rep
print("ok")
""" % REPO
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SK_PACK_SIMD=simd))
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr[-800:]


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


@pytest.mark.parametrize("seed", range(60))
def test_reader_grammar_soup_vs_oracle(seed, tmp_path):
    """files thrown together from header, sequence, '+', quality and blank lines in every order -- equal and unequal
    quality lengths, wrapped sequences, CR line ends, '@' and '>' opening quality lines, a missing last newline,
    plain and gzipped, also cut short -- must give the records the oracle's reader gives (which is pinned to the
    reference's kseq by test_oracle_golden.py).  Covers the parser's whole-record shortcut next to its general path."""
    import gzip
    rng = random.Random(9000 + seed)
    nl = rng.choice([b"\n", b"\n", b"\r\n"])
    strain = _synth.rand_dna(rng, 3000)
    out = []
    for _ in range(rng.randrange(1, 400)):
        kind = rng.random()
        n = rng.choice([0, 1, 2, 30, 31, 40, 150, 151])
        a = rng.randrange(0, len(strain) - n)
        seq = strain[a:a + n]                              # (pieces of the strain: which bytes count as sequence shows in the table)
        if seed % 3 == 0:                                  # every third soup is mostly one-line FASTA: the parser's other whole-record shortcut
            kind = 0.6 if kind < 0.6 else kind
        if kind < 0.55:                                    # a well-formed four-line record (now and then not quite)
            q = bytes(rng.choice(b"FFFF:,#@>+I") for _ in range(n if rng.random() < 0.9 else rng.choice([0, 1, max(0, n - 1), n + 1])))
            out += [b"@r%d some text" % len(out), seq, b"+" + (b"r" if rng.random() < 0.2 else b""), q]
        elif kind < 0.62:                                  # FASTA, the sequence on one line, '>' or '@' header
            out += [(b">" if rng.random() < 0.8 else b"@") + b"o%d len=%d" % (len(out), n), seq]
        elif kind < 0.7:                                   # FASTA, wrapped
            out += [b">c%d" % len(out)] + [seq[i:i + 60] for i in range(0, n, 60)]
        elif kind < 0.8:                                   # FASTQ with wrapped sequence and quality
            q = b"I" * n
            out += [b"@w%d" % len(out)] + [seq[i:i + 50] for i in range(0, n, 50)] + [b"+"] + [q[i:i + 70] for i in range(0, n, 70)]
        else:                                              # loose lines
            out.append(rng.choice([b"", b"+", b"@", b">", b"\r", seq, b"@" + seq, b"+" + seq, b" ", b"\t@x"]))
    text = nl.join(out) + (nl if rng.random() < 0.8 else b"")
    if rng.random() < 0.3:
        text = text[:rng.randrange(len(text) + 1)]
    f = tmp_path / ("soup.fq.gz" if seed % 2 else "soup.fq")
    f.write_bytes(gzip.compress(text, 6, mtime=0) if seed % 2 else text)
    data, nrec, _st = _oracle.decode_file(str(f))
    want = [r for r in data.split(b"\n")[:-1] if len(r) >= 31]
    for chunk_bytes in (1 << 20, 4096):
        chunks, got_nrec, bases = sk.decode_file(str(f), chunk_bytes=chunk_bytes)
        got = b"".join(chunks).split(b"\n")[:-1]
        if chunk_bytes == 1 << 20:
            assert got == want, seed
        assert got_nrec == nrec and bases == len(data) - nrec, seed
    # and the oracle's reader against the reference's own on the same soup, through the two programs
    ref, orc = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count"), os.path.join(REPO, "oracle", "kso_oracle")
    if os.path.exists(ref) and os.path.exists(orc):
        (tmp_path / "strain.fa").write_bytes(b">s\n" + strain + b"\n")
        (tmp_path / "A.txt").write_text(str(f) + "\n")
        argv = ["-r", str(tmp_path / "strain.fa"), "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "A.txt")]
        a, b = (subprocess.run([exe] + argv, capture_output=True, timeout=60) for exe in (ref, orc))
        assert a.returncode == 0 and len(a.stdout.splitlines()) > 2000, seed          # (a table, not the usage text)
        assert (a.returncode, a.stdout) == (b.returncode, b.stdout), seed


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


def test_cpu_budget_follows_quota_and_local_ranks(tmp_path):
    """sk_cpus.h: the default thread budget is the online CPUs cut down to the cgroup quota, shared out among the
    processes of a one-process-per-GPU start"""
    src = tmp_path / "b.c"
    src.write_text('#include "%s"\nint main(void) { printf("%%ld\\n", sk_cpu_budget()); return 0; }\n'
                   % os.path.join(REPO, "strainer2_amd", "csrc", "sk_cpus.h"))
    exe = str(tmp_path / "b")
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", str(src), "-o", exe], check=True)
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE")}
    alone = int(subprocess.run([exe], env=env, capture_output=True, check=True).stdout)
    want = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            want = min(want, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    assert 1 <= alone <= os.cpu_count() and alone == min(os.cpu_count(), want) or alone == os.cpu_count()
    shared = int(subprocess.run([exe], env=dict(env, LOCAL_WORLD_SIZE="4"), capture_output=True, check=True).stdout)
    assert shared == max(1, alone // 4)
    assert int(subprocess.run([exe], env=dict(env, OMPI_COMM_WORLD_LOCAL_SIZE="1000"), capture_output=True, check=True).stdout) == 1


def test_bench_reports_measured_traffic_only_for_the_kernel_it_was_measured_on(repo, tmp_path, monkeypatch):
    """bench.py takes roofline.traffic from profiles/traffic.json only if that file carries the sha256 of the current
    sk_device.hip (tools/save_profile.py stamps it) and belongs to the workload; otherwise null with the reason"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    args = types.SimpleNamespace(reads=10_000_000, read_len=150, strain_bp=5_000_000, hit_frac=0.02)
    tj = json.load(open(os.path.join(repo, "profiles", "traffic.json")))
    got, why = b.measured_traffic(args, "sk_scan_grid")
    if tj.get("sk_device_hip_sha256") == b.device_source_sha():
        assert got == tj["hbm_bytes_per_launch"] and got > 1.5e9
    else:
        assert got is None and "stale" in why
    args.reads = 123
    assert b.measured_traffic(args, "sk_scan_grid")[0] is None
    # the committed facts of the reference's run: present for the timed workload, for all eight ranks
    args.reads = 10_000_000
    f = b.load_cfg2_facts(args)
    assert f is not None and len(f["ranks"]) == 8 and f["ranks"][0]["sum"] == 21279803
    args.hit_frac = 0.1
    assert b.load_cfg2_facts(args) is None
