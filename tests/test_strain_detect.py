"""strain_detect: oracle vs the goldens of the real binary (CPU), and the GPU
program vs the same goldens (-m gpu).  The -o file is compared DECOMPRESSED."""
import gzip
import hashlib
import json
import os
import random
import subprocess

import pytest

import _oracle
import _synth
import strainer2_amd as sk

SD_CASES = ["batch", "cli_se", "cli_pe", "cli_pei", "cli_default", "background", "err_missing", "err_type",
            "err_pe_one_file", "err_b_and_B", "err_no_inf", "err_no_read1"]


def _case(golden, name):
    d = os.path.join(golden, "sd_cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    out = open(os.path.join(d, "expected.stdout"), "rb").read()
    err = open(os.path.join(d, "expected.stderr"), "rb").read()
    hp = os.path.join(d, "expected.hits")
    hits = open(hp, "rb").read() if os.path.exists(hp) else None
    return d, meta, out, err, hits


def _run(runner, d, meta, tmp_path):
    argv = list(meta["argv"])
    of = None
    if "-o" in argv:
        of = str(tmp_path / "out.gz")
        argv[argv.index("-o") + 1] = of
    p = runner(argv, d)
    hits = None
    if of and os.path.exists(of):
        try:
            with gzip.open(of, "rb") as f:
                hits = f.read()
        except EOFError:
            hits = b""
        os.remove(of)
    return p, hits


@pytest.mark.parametrize("name", SD_CASES)
def test_sd_oracle_matches_golden(golden, name, tmp_path):
    d, meta, out, err, hits = _case(golden, name)
    p, got = _run(_oracle.run_sd_oracle_cli, d, meta, tmp_path)
    assert p.returncode == meta["returncode"]
    assert p.stdout == out
    assert p.stderr == err
    if meta["returncode"] == 0:
        assert got == hits


def test_sd_oracle_bundled_step3_md5(golden, tmp_path):
    """reference test/example.sh step 3: 1,122 hit lines + 8 trailer lines, md5 e1799e70..."""
    b = os.path.join(golden, "bundled")
    facts = json.load(open(os.path.join(b, "step3_facts.json")))
    p, got = _run(_oracle.run_sd_oracle_cli, b, facts, tmp_path)
    assert p.returncode == 0 and p.stdout.decode() == facts["stdout"] and p.stderr.decode() == facts["stderr"]
    assert hashlib.md5(got).hexdigest() == facts["hits_md5"] == "e1799e705d4f693240573da32540efcc"
    assert got == open(os.path.join(b, "step3_expected.hits"), "rb").read()


def _gpu_runner(argv, cwd):
    return subprocess.run([sk.cli_path("strain_detect")] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SD_CASES)
def test_sd_program_matches_reference_golden(golden, name, tmp_path):
    d, meta, out, err, hits = _case(golden, name)
    p, got = _run(_gpu_runner, d, meta, tmp_path)
    assert p.returncode == meta["returncode"]
    assert p.stdout == out
    assert p.stderr == err
    if meta["returncode"] == 0:
        assert got == hits


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["batch", "cli_pe", "background", "err_no_inf"])
def test_sd_program_with_the_key_set_built_on_the_host(golden, name, tmp_path, monkeypatch):
    """Round 3: by default the key set of a strain is built ON THE DEVICE from its text (sk_table_build_from_text: keys, first
    occurrences, rows numbered along the text, rank map, filters, column 0) -- the goldens above run that way.  SK_SD_HOST_KEYSET=1
    keeps the host's builder (skh_keyset_from_file + skh_keyset_load), which strains with byte-string keys still take: same output."""
    monkeypatch.setenv("SK_SD_HOST_KEYSET", "1")
    d, meta, out, err, hits = _case(golden, name)
    p, got = _run(_gpu_runner, d, meta, tmp_path)
    assert p.returncode == meta["returncode"] and p.stdout == out and p.stderr == err
    if meta["returncode"] == 0:
        assert got == hits


@pytest.mark.gpu
def test_table_built_on_the_device_equals_the_hosts(tmp_path):
    """sk_table_build_from_text against the host's key set on a strain with repeats, both strands of a segment, N runs, lower
    case, a record of 30 bases (no window) and one of 31: the same SET of keys (the numbering is by first occurrence along the
    text in both, so the same order too), column 0 all 1, and a scan counts every row alike through either table."""
    import ctypes as C
    from strainer2_amd.native import lib
    rng = random.Random(314)
    a = _synth.rand_dna(rng, 3000)
    recs = [a[:1200] + b"NN" + a[1200:2000].lower() + b"N" + a[100:400], _synth.revcomp(a[500:900]) + a[2000:], a[:30], a[7:38], _synth.rand_dna(rng, 500)]
    fa = tmp_path / "s.fa"
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(recs)))
    host = sk.Keyset.from_stream(b"\n".join(recs) + b"\n", default_val=1, incr=0)
    reads = _synth.fuzz_stream(rng, a, 400, junk=b"Nn", p_junk=0.003, min_len=31, max_len=160)
    with sk.KmerContext(0) as ch, sk.KmerContext(0) as cd:
        ch.load_keyset(host, 6)
        ch.scan_stream(reads, 2)
        # the device's way, through the program's own host step (the record parser + packing) via the C entry point
        dks = sk.native._KeysetStruct()
        lib.skh_keyset_build_on_device.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
        rc = lib.skh_keyset_build_on_device(C.byref(dks), cd._h, os.fsencode(str(fa)), 6, 1)
        assert rc == 0
        n = dks.nrows
        # the keys stay on the device until asked for: none fetched yet; a handful of rows first, then the rest, against the whole export
        assert lib_key(dks, 0) == b"A" * 31
        some = (C.c_uint32 * 5)(n - 1, 0, 17, 17, n // 2)
        lib.skh_keyset_fetch_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        assert lib.skh_keyset_fetch_keys(C.byref(dks), cd._h, some, 5) == 0
        whole = (C.c_uint64 * n)()
        lib.sk_table_export_keys.argtypes = [C.c_void_p, C.c_void_p]
        assert lib.sk_table_export_keys(cd._h, whole) == 0
        assert [dks.packed[r] for r in (n - 1, 0, 17, n // 2)] == [whole[r] for r in (n - 1, 0, 17, n // 2)] and dks.packed[1] == 0
        every = (C.c_uint32 * n)(*range(n))
        assert lib.skh_keyset_fetch_keys(C.byref(dks), cd._h, every, n) == 0
        assert list(dks.packed[:n]) == list(whole)
        dev_keys = [lib_key(dks, r) for r in range(n)]
        assert sorted(dev_keys) == sorted(host.keys()) and n == host.nrows
        cd.scan_stream(reads, 2)
        hc, dc = ch.counts(2), cd.counts(2)
        by_key_h = dict(zip(host.keys(), hc.tolist()))
        by_key_d = dict(zip(dev_keys, dc.tolist()))
        assert by_key_h == by_key_d and sum(by_key_h.values()) > 1000
        assert cd.counts(0).tolist() == [1] * n
        # a few rows of a column set to one value (strain_detect's informative rows): the others keep theirs
        lib.sk_counts_set_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        assert lib.sk_counts_set_rows(cd._h, 0, some, 5, 2) == 0
        want = [1] * n
        for r in (n - 1, 0, 17, n // 2):
            want[r] = 2
        assert cd.counts(0).tolist() == want and cd.counts(2).tolist() == dc.tolist()
        assert lib.sk_counts_set_rows(cd._h, 0, (C.c_uint32 * 1)(n), 1, 2) != 0        # a row the table does not have
        lib.skh_keyset_free(C.byref(dks))


def lib_key(dks, row):
    import ctypes as C
    from strainer2_amd.native import lib
    buf = C.create_string_buffer(32)
    lib.skh_keyset_key(C.byref(dks), row, buf)
    return buf.value


@pytest.mark.gpu
def test_sd_program_bundled_step3_md5(golden, tmp_path):
    b = os.path.join(golden, "bundled")
    facts = json.load(open(os.path.join(b, "step3_facts.json")))
    p, got = _run(_gpu_runner, b, facts, tmp_path)
    assert p.returncode == 0 and p.stdout.decode() == facts["stdout"] and p.stderr.decode() == facts["stderr"]
    assert hashlib.md5(got).hexdigest() == "e1799e705d4f693240573da32540efcc"


@pytest.mark.gpu
def test_tally_batch_vs_python_count():
    """sk_tally_batch directly: per-record tallies equal a brute-force count over the key set."""
    rng = random.Random(99)
    strain = _synth.rand_dna(rng, 4000)
    ks = sk.Keyset.from_stream(strain + b"\n", default_val=1, incr=0)
    keys = set(ks.keys())
    inf_rows = set(range(0, ks.nrows, 5))
    inf_keys = {k for r, k in enumerate(ks.keys()) if r in inf_rows}
    reads = _synth.fuzz_stream(rng, strain, 300, junk=b"NnRYKM-. X*", p_junk=0.01, min_len=0, max_len=200).split(b"\n")[:-1]   # (no U: it can match)
    reads = [r for r in reads if len(r) >= 31]
    stream = b"\n".join(reads) + b"\n"
    starts, off = [], 0
    for r in reads:
        starts.append(off)
        off += len(r) + 1
    with sk.KmerContext(0) as ctx:
        ctx.load_keyset(ks, 6)
        import numpy as np
        t = np.ones(ks.nrows, dtype=np.uint32)
        t[sorted(inf_rows)] = 2
        ctx.set_counts(0, t)
        tally, hits = ctx.tally_batch(stream, starts, 0, 2)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i, r in enumerate(reads):
        u = r.upper()
        h = n = 0
        for j in range(len(u) - 30):
            w = u[j:j + 31]
            if set(w) - set(b"ACGT"):
                continue
            c = max(w, w.translate(comp)[::-1])
            if c in keys:
                h += 1
                n += c in inf_keys
        assert (int(tally[i, 0]), int(tally[i, 1])) == (h, n), i
    assert len(hits) == int(tally[:, 1].sum()) > 0


def _oracle_runs(tmp_path, n):
    """the CPU oracle program on strain 0..n-1 of _make_multi_inputs, eight runs at a time; returns the decompressed outfiles"""
    from concurrent.futures import ThreadPoolExecutor

    def one(s):
        ora = _oracle.run_sd_oracle_cli(["-r", str(tmp_path / f"s{s}.fa"), "-a", str(tmp_path / f"s{s}.inf.gz"), "-B", str(tmp_path / "B.txt"),
                                         "-o", str(tmp_path / f"oracle{s}.gz")], str(tmp_path))
        assert ora.returncode == 0
        return gzip.open(tmp_path / f"oracle{s}.gz", "rb").read()
    with ThreadPoolExecutor(max_workers=8) as ex:
        return list(ex.map(one, range(n)))


def _canon(k):
    r = _synth.revcomp(k)
    return k if k >= r else r


def _make_multi_inputs(tmp_path, nstrains=3, strain_len=60_000, nreads=320_000, related=False):
    """nstrains synthetic strains with their informative lists, and a -B list with an SE (.gz), a PE pair and
    an interleaved file whose decoded size spans more than one 32 MiB chunk; short reads mixed in.  related: the
    strains are diverged copies of three ancestors (0.5-3 % substitutions, every seventh an exact copy): most k-mers
    are shared by several strains, and a k-mer informative in one strain is plain in another."""
    rng = random.Random(4242)
    if related:
        anc = [_synth.rand_dna(rng, strain_len) for _ in range(3)]
        strains = []
        for s in range(nstrains):
            g = bytearray(anc[s % 3])
            rate = 0.0 if s % 7 == 6 else rng.choice([0.005, 0.01, 0.03])
            for i in range(strain_len):
                if rng.random() < rate:
                    g[i] = rng.choice(b"ACGT")
            strains.append(bytes(g) if s % 5 else _synth.revcomp(bytes(g)))
    else:
        strains = [_synth.rand_dna(rng, strain_len) for _ in range(nstrains)]
    lines = []
    for s, g in enumerate(strains):
        (tmp_path / f"s{s}.fa").write_bytes(b">s%d\n" % s + g + b"\n")
        kms = sorted({_canon(g[i:i + 31]) for i in range(s % 13 if related else 0, strain_len - 31, 17)})
        with gzip.open(tmp_path / f"s{s}.inf.gz", "wb") as f:
            f.write(b"#informative\n" + b"\n".join(kms) + b"\n")
        lines.append(f"{tmp_path}/s{s}.fa\t{tmp_path}/s{s}.inf.gz\t{tmp_path}/multi{s}.gz\n")
    (tmp_path / "strains.txt").write_text("# genome\tinformative\tout\n" + "".join(lines))

    def reads(n, seed):
        r = random.Random(seed)
        out = []
        for i in range(n):
            if r.random() < 0.03:
                g = strains[r.randrange(nstrains)]
                a = r.randrange(0, strain_len - 150)
                rd = g[a:a + 150]
                if r.random() < 0.5:
                    rd = _synth.revcomp(rd)
            else:
                rd = _synth.rand_dna(r, 150)
            if r.random() < 0.01:
                rd = rd[:r.randrange(0, 31)]                  # shorter than k: inherits the previous read's tallies
            out.append(rd)
        return out

    def fasta(rs):
        return b"".join(b">r%d\n%s\n" % (i, x) for i, x in enumerate(rs))

    with gzip.open(tmp_path / "se.fa.gz", "wb", compresslevel=1) as f:
        f.write(fasta(reads(nreads, 1)))
    (tmp_path / "pe_1.fa").write_bytes(fasta(reads(60_000, 2)))
    (tmp_path / "pe_2.fa").write_bytes(fasta(reads(60_000, 3)))
    (tmp_path / "il.fa").write_bytes(fasta(reads(50_001, 4)))          # odd count: the last mate is missing
    (tmp_path / "B.txt").write_text(f"SE\t{tmp_path}/se.fa.gz\n#comment\nPE\t{tmp_path}/pe_1.fa\t{tmp_path}/pe_2.fa\n"
                                    f"XX\tnope\nPEI\t{tmp_path}/il.fa\n")
    return nstrains


@pytest.mark.gpu
def test_sd_many_strains_in_one_pass_equal_separate_runs(tmp_path):
    """-S <list>: every strain's output is byte-identical (decompressed) to a separate run with its -r/-a/-o,
    and one of the separate runs is checked against the oracle.  With the union table (one scan per batch for all
    strains, the default) and strain by strain (SK_SD_NO_UNION=1)."""
    n = _make_multi_inputs(tmp_path)
    exe = sk.cli_path("strain_detect")
    single = []
    for s in range(n):
        one = subprocess.run([exe, "-r", str(tmp_path / f"s{s}.fa"), "-a", str(tmp_path / f"s{s}.inf.gz"), "-B", str(tmp_path / "B.txt"),
                              "-o", str(tmp_path / f"single{s}.gz")], capture_output=True)
        assert one.returncode == 0
        assert one.stdout == b"unknown file type skipping line (#comment)\nunknown file type skipping line (XX)\n"
        single.append(gzip.open(tmp_path / f"single{s}.gz", "rb").read())
        assert single[s].count(b"\n") > 1000
    ora = _oracle.run_sd_oracle_cli(["-r", str(tmp_path / "s1.fa"), "-a", str(tmp_path / "s1.inf.gz"), "-B", str(tmp_path / "B.txt"),
                                     "-o", str(tmp_path / "oracle1.gz")], str(tmp_path))
    assert ora.returncode == 0
    assert gzip.open(tmp_path / "oracle1.gz", "rb").read() == single[1]
    for union in (True, False):
        env = dict(os.environ, SK_SD_TIMING="1")
        if not union:
            env["SK_SD_NO_UNION"] = "1"
        multi = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", str(tmp_path / "B.txt"), "--coverage-depth"], capture_output=True, env=env)
        assert (b"union table(s) for 3 strains" in multi.stderr) == union, multi.stderr.decode()[-800:]
        assert multi.returncode == 0, multi.stderr.decode()[-500:]
        assert multi.stdout == b"unknown file type skipping line (#comment)\nunknown file type skipping line (XX)\n"
        for s in range(n):
            assert gzip.open(tmp_path / f"multi{s}.gz", "rb").read() == single[s], (union, s)
            cov = subprocess.run([sk.cli_path("coverage_depth"), "-k", str(tmp_path / f"multi{s}.gz")], capture_output=True)
            assert cov.returncode == 0 and cov.stdout == open(tmp_path / f"multi{s}.gz.coverage_depth", "rb").read()
            os.remove(tmp_path / f"multi{s}.gz")


@pytest.mark.gpu
def test_sd_many_strains_golden_strain_plus_another(golden, tmp_path):
    """two different strains against the `batch` golden's metagenome list: the golden strain's file equals the
    reference's output, the other equals its own separate run."""
    d = os.path.join(golden, "sd_cases", "batch")
    other = os.path.join(golden, "sd_cases", "background")
    (tmp_path / "strains.txt").write_text(f"strain.fa\tinf.txt.gz\t{tmp_path}/a.gz\n{other}/strain.fa\t{other}/inf.txt\t{tmp_path}/b.gz\n")
    exe = sk.cli_path("strain_detect")
    p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", "B.txt"], cwd=d, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert gzip.open(tmp_path / "a.gz", "rb").read() == open(os.path.join(d, "expected.hits"), "rb").read()
    q = subprocess.run([exe, "-r", f"{other}/strain.fa", "-a", f"{other}/inf.txt", "-B", "B.txt", "-o", str(tmp_path / "b1.gz")], cwd=d, capture_output=True)
    assert q.returncode == 0
    assert gzip.open(tmp_path / "b.gz", "rb").read() == gzip.open(tmp_path / "b1.gz", "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("union", [True, False])
def test_sd_one_decoder_several_devices(golden, tmp_path, union):
    """SK_DEVICES=0,0,0: ONE process, ONE decode pipeline, three logical devices (all of them this box's one card -- on a node
    they are three GPUs): the strains go to the devices in groups (SK_SD_GROUP=2: five strains = three groups = three union
    tables, one per device), every decoded chunk is uploaded to every device, each device tallies it against its own strains.
    Every strain's file equals the reference's output for the single strain (golden `batch`), through the union tables and
    strain by strain; small chunks so that the next chunk's prefetch to all devices is in play."""
    d = os.path.join(golden, "sd_cases", "batch")
    exe = sk.cli_path("strain_detect")
    with open(tmp_path / "strains.txt", "w") as f:
        for s in range(5):
            f.write(f"strain.fa\tinf.txt.gz\t{tmp_path}/m{s}.gz\n")
    want = open(os.path.join(d, "expected.hits"), "rb").read()
    env = dict(os.environ, SK_DEVICES="0,0,0", SK_SD_GROUP="2", SK_SD_TIMING="1", SK_SD_CHUNK_BYTES="3000")
    if not union:
        env["SK_SD_NO_UNION"] = "1"
    p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", "B.txt"], cwd=d, env=env, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert b"5 strains on 3 devices, one decode pipeline" in p.stderr
    assert (b"3 union table(s) for 5 strains" in p.stderr) == union
    for s in range(5):
        assert gzip.open(tmp_path / f"m{s}.gz", "rb").read() == want, s


@pytest.mark.gpu
def test_sd_strain_list_is_dealt_to_ranks(golden, tmp_path):
    """-S with RANK/WORLD_SIZE: strain i of the list belongs to rank i % world (no collective); each rank writes
    only its own outfiles, and they equal the single-process ones"""
    d = os.path.join(golden, "sd_cases", "batch")
    exe = sk.cli_path("strain_detect")
    with open(tmp_path / "strains.txt", "w") as f:
        for s in range(3):
            f.write(f"strain.fa\tinf.txt.gz\t{tmp_path}/r{s}.gz\n")
    want = open(os.path.join(d, "expected.hits"), "rb").read()
    for rank in (0, 1):
        env = dict(os.environ, SK_WORLD_SIZE="2", SK_RANK=str(rank), SK_LOCAL_RANK="0")
        p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", "B.txt"], cwd=d, env=env, capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[-500:]
    assert sorted(x for x in os.listdir(tmp_path) if x.startswith("r")) == ["r0.gz", "r1.gz", "r2.gz"]
    for s in range(3):
        assert gzip.open(tmp_path / f"r{s}.gz", "rb").read() == want
    # a world of two where one rank has nothing to do
    (tmp_path / "one.txt").write_text(f"strain.fa\tinf.txt.gz\t{tmp_path}/only.gz\n")
    p = subprocess.run([exe, "-S", str(tmp_path / "one.txt"), "-B", "B.txt"], cwd=d, capture_output=True,
                       env=dict(os.environ, SK_WORLD_SIZE="2", SK_RANK="1"))
    assert p.returncode == 0 and not os.path.exists(tmp_path / "only.gz")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["batch", "cli_pe", "cli_pei", "background"])
def test_sd_program_tiny_chunks(golden, name, tmp_path, monkeypatch):
    """the same goldens with chunks of a few reads (SK_SD_CHUNK_BYTES): mates in different chunks, state carried
    across chunk boundaries, many small tally launches"""
    monkeypatch.setenv("SK_SD_CHUNK_BYTES", "500")
    d, meta, out, err, hits = _case(golden, name)
    p, got = _run(_gpu_runner, d, meta, tmp_path)
    assert p.returncode == meta["returncode"] == 0
    assert p.stdout == out and p.stderr == err
    assert got == hits


@pytest.mark.gpu
@pytest.mark.parametrize("name,chunk", [("batch", "500"), ("cli_pe", "3000"), ("cli_pei", "0"), ("background", "0")])
def test_sd_program_with_packed_chunks(golden, name, chunk, tmp_path, monkeypatch):
    """SK_SD_PACK=1: the parser threads hand their chunks on PACKED (sk_pack_stream: 6 bytes per 16 bases; sk_batch_fill_packed ->
    sk_scan_grid<TALLY[,UNION],PACKED>) -- the goldens' result files, stdout and stderr must not change (opt-in: on a host with 16
    CPUs per card the packing costs more than the link saves)"""
    monkeypatch.setenv("SK_SD_PACK", "1")
    if chunk != "0":
        monkeypatch.setenv("SK_SD_CHUNK_BYTES", chunk)
    d, meta, out, err, hits = _case(golden, name)
    p, got = _run(_gpu_runner, d, meta, tmp_path)
    assert p.returncode == meta["returncode"] == 0
    assert p.stdout == out and p.stderr == err
    assert got == hits


@pytest.mark.gpu
def test_sd_strain_list_with_background_column(golden, tmp_path):
    """-S with the optional 4th column (-g list): the background filter runs per strain on its worker thread and
    its messages come out in list order; same result as the single-strain golden"""
    d, meta, out, err, hits = _case(golden, "background")
    (tmp_path / "strains.txt").write_text(f"strain.fa\tinf.txt\t{tmp_path}/a.gz\tbg.txt\nstrain.fa\tinf.txt\t{tmp_path}/b.gz\tbg.txt\n")
    p = subprocess.run([sk.cli_path("strain_detect"), "-S", str(tmp_path / "strains.txt"), "-B", "B.txt"], cwd=d, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert p.stdout == out + out and p.stderr == err + err
    for f in ("a.gz", "b.gz"):
        assert gzip.open(tmp_path / f, "rb").read() == hits


@pytest.mark.gpu
def test_sd_32_strains_against_32_oracle_runs(tmp_path):
    """cfg 5 in miniature (one GPU's share of 256 strains is 32): 32 small strains resident at once, ONE pass over the
    metagenomes (each chunk uploaded once, 32 launches, sparse tallies back), every strain's hit list byte-identical
    (decompressed) to a separate run of the CPU oracle program on that strain (src/strain_detect.c:387-663)."""
    n = _make_multi_inputs(tmp_path, nstrains=32, strain_len=20_000, nreads=60_000)
    exe = sk.cli_path("strain_detect")
    multi = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", str(tmp_path / "B.txt")], capture_output=True)
    assert multi.returncode == 0, multi.stderr.decode()[-500:]
    total = 0
    for s, want in enumerate(_oracle_runs(tmp_path, n)):
        assert gzip.open(tmp_path / f"multi{s}.gz", "rb").read() == want, s
        total += want.count(b"\n")
    assert total > 32 * 50


@pytest.mark.gpu
def test_sd_related_strains_in_two_unions_against_oracle_runs(tmp_path):
    """40 strains that share most of their k-mers (diverged and exact copies of three ancestors, some on the other
    strand), i.e. two union tables (32 + 8): a shared k-mer has one slot in a union but counts for every strain
    that holds it, and is informative only where that strain's list says so.  Every strain's hit list byte-identical
    to the CPU oracle program's run on that strain alone (src/strain_detect.c:387-663)."""
    n = _make_multi_inputs(tmp_path, nstrains=40, strain_len=12_000, nreads=40_000, related=True)
    exe = sk.cli_path("strain_detect")
    multi = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", str(tmp_path / "B.txt")], capture_output=True,
                           env=dict(os.environ, SK_SD_TIMING="1"))
    assert multi.returncode == 0, multi.stderr.decode()[-500:]
    assert b"2 union table(s) for 40 strains" in multi.stderr
    total = 0
    for s, want in enumerate(_oracle_runs(tmp_path, n)):
        assert gzip.open(tmp_path / f"multi{s}.gz", "rb").read() == want, s
        total += want.count(b"\n")
    assert total > 40 * 200


@pytest.mark.gpu
def test_sd_union_leaves_a_giant_record_to_the_members(tmp_path):
    """a chunk of 64 MiB or more (one 70 Mbase record: a chunk never cuts a record) does not fit the union's log packing:
    that chunk is tallied strain by strain, the others through the union -- the outfiles are what SK_SD_NO_UNION=1 writes"""
    rng = random.Random(99)
    strains = [_synth.rand_dna(rng, 20_000) for _ in range(3)]
    lines = []
    for s, g in enumerate(strains):
        (tmp_path / f"s{s}.fa").write_bytes(b">s%d\n" % s + g + b"\n")
        kms = sorted({_canon(g[i:i + 31]) for i in range(0, len(g) - 31, 23)})
        (tmp_path / f"s{s}.inf").write_bytes(b"\n".join(kms) + b"\n")
        lines.append(f"{tmp_path}/s{s}.fa\t{tmp_path}/s{s}.inf\t{tmp_path}/out{s}.gz\n")
    (tmp_path / "strains.txt").write_text("".join(lines))
    import numpy as np
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(5).integers(0, 4, 70_000_000)].tobytes()
    big = big[:1_000_000] + strains[1][:5000] + big[1_005_000:40_000_000] + _synth.revcomp(strains[2][100:9000]) + big[40_008_900:]
    with open(tmp_path / "reads.fa", "wb") as f:
        f.write(b">r0\n" + strains[0][50:200] + b"\n>giant\n" + big + b"\n>r2\n" + strains[1][300:450] + b"\n")
    exe = sk.cli_path("strain_detect")
    outs = {}
    for union in (True, False):
        env = dict(os.environ, SK_SD_TIMING="1")
        if not union:
            env["SK_SD_NO_UNION"] = "1"
        p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-b", str(tmp_path / "reads.fa"), "-t", "SE"], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        outs[union] = [gzip.open(tmp_path / f"out{s}.gz", "rb").read() for s in range(3)]
    assert outs[True] == outs[False]
    assert sum(o.count(b"\n") for o in outs[True]) > 300
