"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle and the
golden vectors.  Bit-exact (integer work): every comparison is equality."""
import hashlib
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import _oracle
import _synth
import strainer2_amd as sk
from strainer2_amd import synth

pytestmark = pytest.mark.gpu

CASES = ["mixed", "drug", "iupac_strain", "truncated_fastq", "missing_in_list", "missing_flag", "contig30",
         "progress_missing", "skip_after_missing"]


@pytest.fixture(scope="module")
def ctx():
    c = sk.KmerContext(0)
    yield c
    c.close()


def _golden_case(golden, name):
    d = os.path.join(golden, "cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    return d, meta, open(os.path.join(d, "expected.stdout"), "rb").read(), open(os.path.join(d, "expected.stderr"), "rb").read()


@pytest.mark.parametrize("pack", ["1", "2"])
@pytest.mark.parametrize("name", CASES)
def test_program_matches_reference_golden(golden, name, tmp_path, pack):
    """bin/kmer_scrub_count vs the bytes the unmodified reference printed (stdout, stderr, status, progress); SK_LIST_PACK=2: with
    every chunk of every list uploaded in the host-packed form (chunks with IUPAC letters, U or CR fall back to bytes by themselves)"""
    d, meta, out, err = _golden_case(golden, name)
    argv = list(meta["argv"])
    if "-p" in argv:
        argv[argv.index("-p") + 1] = str(tmp_path / "progress")
    p = subprocess.run([sk.cli_path()] + argv, cwd=d, capture_output=True, env=dict(os.environ, SK_LIST_PACK=pack))
    assert p.returncode == meta["returncode"]
    assert p.stdout == out
    assert p.stderr == err
    if meta["progress_col1"] is not None:
        with open(tmp_path / "progress") as f:
            assert [ln.split("\t")[0].rstrip("\n") for ln in f] == meta["progress_col1"]


@pytest.mark.parametrize("sync", ["blocking", "yield"])
def test_program_with_waits_that_sleep(golden, sync):
    """SK_SYNC: the process's waits for the device sleep (hipDeviceScheduleBlockingSync, blocking events) or yield instead of spinning --
    an option for hosts that are short of CPUs; the table is the same"""
    for name in ("mixed", "drug"):
        d, meta, out, err = _golden_case(golden, name)
        p = subprocess.run([sk.cli_path()] + list(meta["argv"]), cwd=d, capture_output=True, env=dict(os.environ, SK_SYNC=sync, SK_THREADS="3", SK_CHUNK_BYTES="8192"))
        assert p.returncode == meta["returncode"] and p.stdout == out and p.stderr == err


def test_program_through_rccl_path_single_rank(golden, tmp_path):
    """SK_FORCE_COMM=1: the one-process-per-GPU code path (RCCL unique-id rendezvous, failure agreement,
    all-reduce of the counter block, rank 0 prints) with a world of one; output must not change."""
    d, meta, out, err = _golden_case(golden, "drug")
    argv = list(meta["argv"])
    argv[argv.index("-p") + 1] = str(tmp_path / "progress")
    env = dict(os.environ, SK_FORCE_COMM="1", SK_RCCL_ID_FILE=str(tmp_path / "rccl_id"))
    p = subprocess.run([sk.cli_path()] + argv, cwd=d, capture_output=True, env=env)
    assert p.returncode == 0, p.stderr
    assert p.stdout == out
    assert p.stderr.endswith(err)                     # (RCCL prints its version banner first)
    # a rank that cannot read one of its files makes every rank stop with status 1 and no table
    d2, meta2, _o, err2 = _golden_case(golden, "missing_in_list")
    p = subprocess.run([sk.cli_path()] + meta2["argv"], cwd=d2, capture_output=True, env=env)
    assert p.returncode == 1 and p.stdout == b"" and p.stderr.endswith(err2)


def test_program_short_contig_divergence(golden):
    """The reference crashes (SIGSEGV) on a strain record shorter than k-1; we skip it, say so on
    stderr, and otherwise produce what the oracle produces with the record skipped."""
    d, meta, _out, _err = _golden_case(golden, "short_contig")
    assert meta["returncode"] == -11
    p = subprocess.run([sk.cli_path()] + meta["argv"], cwd=d, capture_output=True)
    assert p.returncode == 0
    assert b"skipped 1 reference record(s) shorter than 30 bases" in p.stderr
    t = _oracle.OracleTable()
    assert t.build_file(os.path.join(d, "strain.fa"), short_policy=1) == 0
    t.scan_file(os.path.join(d, "strain.fa"), 1)
    t.scan_file(os.path.join(d, "strain.fa"), 2)
    keys, counts = t.rows()
    want = b"#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n" + b"".join(
        k + b"\t%d\t%d\t%d\n" % tuple(int(x) for x in c[:3]) for k, c in zip(keys, counts))
    assert p.stdout == want


def test_program_bundled_example_md5(golden, tmp_path):
    """cfg 1 through the GPU path: reference test/example.sh step 1, md5 of the 254 MB TSV."""
    b = os.path.join(golden, "bundled")
    facts = json.load(open(os.path.join(b, "step1_facts.json")))
    out = tmp_path / "step1.tsv"
    with open(out, "wb") as f:
        p = subprocess.run([sk.cli_path()] + facts["argv"] + ["-p", str(tmp_path / "prog")], cwd=b, stdout=f,
                           stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stderr == b""
    h = hashlib.md5()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert os.path.getsize(out) == facts["stdout_bytes"]
    assert h.hexdigest() == facts["stdout_md5"] == "75989a9bc31ef0b6f53a5112a60920bd"


def test_program_table_after_two_expansions_md5(golden, tmp_path):
    """a 9.2 Mbp strain: the reference's hash table grows twice (8 M -> 16 M -> 32 M slots, src/BIO_hash.c:129-139), and the row
    order of its TSV is what that leaves.  The program's whole stdout (350 MB) against the md5 of the UNMODIFIED reference's
    (tests/golden/two_expansions_facts.json, made by tests/golden/make_two_expansions_facts.py)."""
    path, facts = _synth.two_expansions_strain(golden, tmp_path)
    if path is None:
        pytest.skip("this numpy's generator draws another strain than the one the facts were made from")
    for n in ("A.txt", "B.txt"):
        (tmp_path / n).write_text("")
    out = tmp_path / "table.tsv"
    with open(out, "wb") as f:
        p = subprocess.run([sk.cli_path(), "-r", path, "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")], stdout=f, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stderr == b""
    h = hashlib.md5()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    assert os.path.getsize(out) == facts["stdout_bytes"]
    assert h.hexdigest() == facts["stdout_md5"]
    os.unlink(out)


@pytest.mark.parametrize("seed,junk", [(1, 0.0), (2, 0.01), (3, 0.05)])
def test_scan_stream_fuzz_vs_oracle(ctx, seed, junk):
    """Random streams with junk bytes / case / N / U / IUPAC, strain with IUPAC letters too."""
    rng = random.Random(seed)
    strain = bytearray(_synth.rand_dna(rng, 6000))
    if junk:
        for ch in b"RYKMSWU":
            strain[rng.randrange(len(strain))] = ch
    sstream = bytes(strain[:3000]) + b"\n" + bytes(strain[3000:]).lower() + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    ctx.load_keyset(ks, 4)
    for col in (1, 2, 3):
        data = _synth.fuzz_stream(rng, bytes(strain), 400, p_junk=junk, min_len=0, max_len=260)
        ctx.scan_stream(data, col)
        t.scan_stream(data, col)
    okeys, ocounts = t.rows()
    assert ks.keys() == okeys
    for col in range(4):
        assert np.array_equal(ctx.counts(col), ocounts[:, col]), col
    assert ocounts[:, 1:].sum() > 1000          # the test is not vacuous


@pytest.mark.parametrize("text_stage,pipeline", [(1, 1), (0, 1), (1, 2), (0, 2)])
def test_both_stage2_paths_vs_oracle(text_stage, pipeline):
    """stage 2 with the strain's text (seed and verify, default) and without it (every window probed on its own),
    on a stream that crosses several tiles with junk, short records and strain reads"""
    rng = random.Random(77 + text_stage)
    strain = _synth.rand_dna(rng, 50_000)
    sstream = strain + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    data = _synth.fuzz_stream(rng, strain, 3000, p_junk=0.003, min_len=0, max_len=400)
    assert len(data) > 5 * 32768
    with sk.KmerContext(0) as c:
        c.set_option("text_stage", text_stage)
        c.set_option("pipeline", pipeline)         # 1 = the single kernel, 2 = sk_bin -> sk_lds_probe -> candidates only
        c.load_keyset(ks, 4)
        c.scan_stream(data, 2)
        t.scan_stream(data, 2)
        _, ocounts = t.rows()
        got = c.counts(2)
    assert np.array_equal(got, ocounts[:, 2]) and got.sum() > 10_000


@pytest.mark.parametrize("cap", [0, 3])
def test_byte_string_path_list_and_overflow(cap):
    """windows with IUPAC letters / U are found through the list of chunks that hold such bytes; with the list
    cut to 3 entries it overflows and the byte-string kernel visits every position instead: same counts"""
    rng = random.Random(5150)
    strain = bytearray(_synth.rand_dna(rng, 30_000))
    for ch in b"RYKMSW":
        for _ in range(3):
            strain[rng.randrange(len(strain))] = ch
    sstream = bytes(strain) + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    data = _synth.fuzz_stream(rng, bytes(strain), 1500, p_junk=0.004, min_len=0, max_len=300)
    with sk.KmerContext(0) as c:
        c.set_option("odd_list_cap", cap)
        c.load_keyset(ks, 4)
        c.scan_stream(data, 1)
        t.scan_stream(data, 1)
        _, ocounts = t.rows()
        got = c.counts(1)
    assert np.array_equal(got, ocounts[:, 1]) and got.sum() > 5_000


def test_scan_device_equals_scan_stream_and_tile_edges(ctx):
    """Device-resident entry point; stream lengths around tile (32768) and chunk (16) edges."""
    rng = random.Random(7)
    strain = _synth.rand_dna(rng, 50000)
    ks = sk.Keyset.from_stream(strain + b"\n")
    ctx.load_keyset(ks, 4)
    t = _oracle.OracleTable()
    t.build_stream(strain + b"\n")
    base = (strain * 3)[:140000]
    for n in (0, 1, 30, 31, 32, 47, 32767, 32768, 32769, 32768 + 31, 65536 + 15, 98304, 140000):
        data = base[:n]
        ctx.zero_counts(1)
        ctx.zero_counts(2)
        ctx.scan_stream(data, 1)
        buf = ctx.dev_alloc(max(n, 16))
        if n:
            ctx.dev_upload(buf, np.frombuffer(data, dtype=np.uint8))
        ctx.scan_device(buf, n, 2)
        ctx.sync()
        ctx.dev_free(buf)
        t2 = _oracle.OracleTable()
        t2.build_stream(strain + b"\n")
        t2.scan_stream(data, 1)
        want = t2.rows()[1][:, 1]
        assert np.array_equal(ctx.counts(1), want), n
        assert np.array_equal(ctx.counts(2), want), n


def test_file_scan_and_list_sharding_identity(ctx, golden, tmp_path):
    """Counts are additive: scanning a list as 2 shards (rank/world) sums to the unsharded scan."""
    d = os.path.join(golden, "cases", "mixed")
    ks = sk.Keyset.from_file(os.path.join(d, "strain.fna.gz"))
    ctx.load_keyset(ks, 4)
    lst = tmp_path / "L.txt"
    lst.write_text("\n".join(os.path.join(d, f) for f in ["m1.fasta", "m2.fq.gz", "m3_crlf.fa", "g1.fa", "g2.fa.gz"]) + "\n")
    ctx.scan_list(str(lst), 1)
    ctx.scan_list(str(lst), 2, rank=0, world=2)
    ctx.scan_list(str(lst), 3, rank=1, world=2)
    whole, a, b = ctx.counts(1), ctx.counts(2), ctx.counts(3)
    assert whole.sum() > 0 and a.sum() > 0 and b.sum() > 0
    assert np.array_equal(whole, a + b)


def test_full_size_properties_cfg2(ctx):
    """BASELINE cfg 2 scale (5 Mbp strain; here 1 M reads of it to bound host memory/time) through
    size-independent properties: oracle equality on a sample, linearity, strand symmetry, shard sums."""
    contigs = synth.make_strain()
    sstream = synth.strain_stream(contigs)
    ks = sk.Keyset.from_stream(sstream)
    assert 4_990_000 < ks.nrows <= 5_000_000
    ctx.load_keyset(ks, 4)
    reads, nbases = synth.make_reads(contigs, 1_000_000)
    assert nbases == 150_000_000
    ctx.scan_stream(reads, 1)
    c1 = ctx.counts(1)
    # (a) oracle on the first 20 k reads
    sample = reads[: 20_000 * 151].tobytes()
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    t.scan_stream(sample, 1)
    okeys, ocounts = t.rows()
    ctx.scan_stream(sample, 2)
    assert np.array_equal(ctx.counts(2), ocounts[:, 1])
    assert np.array_equal(ks.first_count(), ocounts[:, 0])
    # (b) linearity: scanning the stream again doubles every count
    ctx.scan_stream(reads, 1)
    assert np.array_equal(ctx.counts(1), 2 * c1)
    # (c) strand symmetry: reverse-complemented reads give the same table
    rc = reads.reshape(-1, 151).copy()
    rc[:, :150] = synth._COMP[rc[:, :150]][:, ::-1]
    ctx.zero_counts(3)
    ctx.scan_stream(rc.reshape(-1), 3)
    assert np.array_equal(ctx.counts(3), c1)
    # (d) shard sums: two halves add up
    ctx.zero_counts(2)
    ctx.zero_counts(3)
    half = 500_000 * 151
    ctx.scan_stream(reads[:half], 2)
    ctx.scan_stream(reads[half:], 3)
    assert np.array_equal(ctx.counts(2) + ctx.counts(3), c1)
    # (e) planted hits are found: ~2 % of reads x up to 120 windows
    assert 1_500_000 < int(c1.sum()) < 2_600_000


def test_strain_against_itself_reproduces_multiplicity(ctx):
    """Property at full key-set size: scanning the strain's own contigs counts every k-mer exactly as
    often as the build phase saw it (column 0), both strands; every window is a hit (100 % pass rate
    through both filter stages and queues)."""
    contigs = synth.make_strain()
    sstream = synth.strain_stream(contigs)
    ks = sk.Keyset.from_stream(sstream)
    ctx.load_keyset(ks, 4)
    ctx.scan_stream(sstream, 1)
    rc = b"\n".join(bytes(synth._COMP[c][::-1]) for c in contigs) + b"\n"
    ctx.scan_stream(rc, 2)
    assert np.array_equal(ctx.counts(1), ks.first_count())
    assert np.array_equal(ctx.counts(2), ks.first_count())
    assert np.array_equal(ctx.counts(0), ks.first_count())


def test_low_complexity_and_repeats_vs_oracle(ctx):
    """Homopolymers, short tandem repeats and long exact repeats: one minimizer for very long runs of
    windows (events of maximal length), heavy key multiplicity, palindromic 16-mers."""
    rng = random.Random(1234)
    unit = _synth.rand_dna(rng, 700)
    strain = (b"A" * 200 + unit + b"ACACACACAC" * 30 + unit + b"T" * 150 + b"GATC" * 60 + _synth.rand_dna(rng, 500) +
              b"AATT" * 40 + unit[::-1] + b"C" * 90)
    sstream = strain + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    ctx.load_keyset(ks, 4)
    reads = [strain[i:i + 180] for i in range(0, len(strain) - 180, 37)]
    reads += [b"A" * 300, b"T" * 77, b"AC" * 100, b"GATC" * 50, _synth.revcomp(strain[150:600]), b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAC" * 4]
    data = b"\n".join(reads) + b"\n"
    ctx.scan_stream(data, 1)
    t.scan_stream(data, 1)
    okeys, ocounts = t.rows()
    assert ks.keys() == okeys
    assert np.array_equal(ctx.counts(0), ocounts[:, 0])
    assert np.array_equal(ctx.counts(1), ocounts[:, 1])
    assert int(ocounts[:, 1].max()) > 20


def test_plain_table_load_without_locality(ctx):
    """sk_table_load (no locality permutation: counters in caller row order, no strain-order key copy)
    must count exactly like the keyset path."""
    rng = random.Random(77)
    strain = _synth.rand_dna(rng, 30000)
    ks = sk.Keyset.from_stream(strain + b"\n")
    data = _synth.fuzz_stream(rng, strain, 1500, junk=b"NnRY-", p_junk=0.005, min_len=20, max_len=250)
    ctx.load_keyset(ks, 4)
    ctx.scan_stream(data, 2)
    want = ctx.counts(2)
    keys = ks.packed()
    order = np.arange(len(keys))
    rng2 = np.random.default_rng(5)
    rng2.shuffle(order)                                   # any row order the caller likes
    ctx.load_table(keys[order], 4)
    ctx.scan_stream(data, 2)
    got = ctx.counts(2)
    assert np.array_equal(got, want[order])
    assert int(want.sum()) > 10000


def test_program_cfg3_shape_vs_oracle_program(tmp_path):
    """BASELINE cfg 3 in miniature through the programs: a strain, an -A list of genomes with long
    contigs (two of them 1 % diverged copies of the strain, one on the other strand), a -B list of
    gz/plain FASTQ read files, a -C list that contains the -r path itself; decoded by the host thread
    pool.  The whole TSV must equal the oracle program's."""
    import gzip
    rng = random.Random(2024)
    contigs = [_synth.rand_dna(rng, 120_000), _synth.rand_dna(rng, 80_000)]
    (tmp_path / "strain.fa").write_bytes(b"".join(b">c%d\n" % i + b"\n".join(c[j:j + 70] for j in range(0, len(c), 70)) + b"\n"
                                                   for i, c in enumerate(contigs)))

    def mutate(seq, rate):
        b = bytearray(seq)
        for i in range(len(b)):
            if rng.random() < rate:
                b[i] = rng.choice(b"ACGT")
        return bytes(b)

    genomes = [mutate(contigs[0] + contigs[1], 0.01), _synth.revcomp(mutate(contigs[0], 0.01)),
               _synth.rand_dna(rng, 200_000), _synth.rand_dna(rng, 150_000) + b"NNNN" + contigs[1][:5000]]
    names = []
    for i, g in enumerate(genomes):
        n = "g%d.fa" % i
        (tmp_path / n).write_bytes(b">g%d\n" % i + b"\n".join(g[j:j + 80] for j in range(0, len(g), 80)) + b"\n")
        names.append(n)
    (tmp_path / "A.txt").write_text("\n".join(names) + "\n")
    whole = contigs[0] + contigs[1]
    bnames = []
    for f in range(4):
        recs = []
        for i in range(4000):
            if rng.random() < 0.3:
                a = rng.randrange(0, len(whole) - 150)
                r = mutate(whole[a:a + 150], 0.005)
                if rng.random() < 0.5:
                    r = _synth.revcomp(r)
            else:
                r = _synth.rand_dna(rng, 150)
            recs.append(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
        n = "reads%d.fq%s" % (f, ".gz" if f % 2 else "")
        data = b"".join(recs)
        (tmp_path / n).write_bytes(gzip.compress(data, 4) if f % 2 else data)
        bnames.append(n)
    (tmp_path / "B.txt").write_text("\n".join(bnames) + "\n")
    (tmp_path / "C.txt").write_text("g1.fa\nstrain.fa\ng0.fa\n")
    argv = ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-C", "C.txt"]
    want = _oracle.run_oracle_cli(argv, str(tmp_path))
    got = subprocess.run([sk.cli_path()] + argv, cwd=str(tmp_path), capture_output=True)
    assert want.returncode == got.returncode == 0
    assert got.stderr == want.stderr == b"skipping strain.fa (identical match)\n"
    assert got.stdout == want.stdout
    cols = np.array([[int(x) for x in ln.split(b"\t")[1:]] for ln in got.stdout.split(b"\n")[1:] if ln])
    assert cols.shape[1] == 4 and (cols.sum(axis=0) > [190_000, 200_000, 100_000, 200_000]).all(), cols.sum(axis=0)


def test_single_launch_beyond_4_gib(ctx, golden):
    """BASELINE configs[1] at full size through sk_scan_device, pinned to the UNMODIFIED reference: the metagenome_count
    column of the rank-0 stream (10 M x 150 bp vs the 5 Mbp strain) must have the sum, the number of non-zero rows and
    the md5 the reference program produced for this very stream (tests/golden/cfg2_facts.json, made by
    tests/golden/make_cfg2_facts.py; src/kmer_scrub_count.c:89-98,134-156).  And one device-resident batch of more than
    2^32 bytes (64-bit positions inside the kernel): the same 1.51 GB stream three times in a row must count exactly
    three times one copy."""
    contigs = synth.make_strain()
    ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
    ctx.load_keyset(ks, 4)
    reads, _ = synth.make_reads(contigs, 10_000_000)
    n = int(reads.size)
    assert 3 * n > 2 ** 32
    buf = ctx.dev_alloc(3 * n + 16)
    for i in range(3):
        ctx.dev_upload(buf, reads, offset=i * n)
    ctx.scan_device(buf, n, 1)
    ctx.scan_device(buf, 3 * n, 2)
    ctx.sync()
    one, three = ctx.counts(1), ctx.counts(2)
    ctx.dev_free(buf)
    facts = json.load(open(os.path.join(golden, "cfg2_facts.json")))
    want = facts["ranks"][0]
    assert facts["reads"] == 10_000_000 and facts["read_len"] == 150 and want["rows"] == ks.nrows
    got = {"sum": int(one.astype(np.uint64).sum()), "nonzero_rows": int(np.count_nonzero(one)), "max": int(one.max()),
           "md5_u32_le": hashlib.md5(one.astype("<u4").tobytes()).hexdigest()}
    assert got == {k: want[k] for k in got}, (got, want)
    assert np.array_equal(three, 3 * one)


def test_scan_pinned_equals_scan_stream(ctx):
    """Zero-copy entry point: chunks of whole records in pinned memory, tickets gate buffer reuse."""
    rng = random.Random(4242)
    strain = _synth.rand_dna(rng, 40000)
    ks = sk.Keyset.from_stream(strain + b"\n")
    ctx.load_keyset(ks, 4)
    data = _synth.fuzz_stream(rng, strain, 3000, junk=b"Nn", p_junk=0.002, min_len=31, max_len=200)
    ctx.scan_stream(data, 1)
    recs = data.split(b"\n")[:-1]
    buf = ctx.pinned_alloc(1 << 16)
    tickets = []
    i = 0
    while i < len(recs):                                  # refill the same small buffer again and again
        n, j = 0, i
        while j < len(recs) and n + len(recs[j]) + 1 <= buf.size:
            n += len(recs[j]) + 1
            j += 1
        if tickets:
            ctx.ticket_wait(tickets[-1])
        chunk = b"\n".join(recs[i:j]) + b"\n"
        buf[:n] = np.frombuffer(chunk, dtype=np.uint8)
        tickets.append(ctx.scan_pinned(buf, n, 2))
        i = j
    ctx.sync()
    assert len(tickets) > 5
    assert np.array_equal(ctx.counts(1), ctx.counts(2)) and int(ctx.counts(1).sum()) > 10000
    ctx.pinned_free(buf)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_packed_batches_count_what_the_byte_batches_count(ctx, seed):
    """the host-side 2-bit pre-pack (sk_pack_stream -> sk_scan_pinned_packed): chunks of whole records, packed on the host -- 6 bytes
    per 16 bases -- must give, count for count, what the same chunks give as bytes and what the oracle counts: strain reads on both
    strands with substitutions, N and n, lower case, reads around k and around the 16-byte grid, ragged chunk ends, a chunk of one
    byte.  A chunk with a byte for the byte-string kernel says so (`odd`) and goes up as bytes."""
    rng = random.Random(seed)
    strain = _synth.rand_dna(rng, rng.choice([20000, 60000]))
    ks = sk.Keyset.from_stream(strain + b"\n")
    ctx.load_keyset(ks, 4)
    data = _synth.fuzz_stream(rng, strain, 4000, junk=b"NnNnacgt", p_junk=0.004, min_len=rng.choice([1, 31]), max_len=rng.choice([64, 200, 700]))
    ctx.scan_stream(data, 1)
    recs = data.split(b"\n")[:-1]
    cap = rng.choice([1 << 12, 1 << 16, 1 << 20])
    raw = ctx.pinned_alloc(cap)
    pk = [ctx.pinned_alloc(6 * ((cap + 15) // 16)) for _ in range(2)]
    tickets = [None, None]
    i = k = 0
    npacked = 0
    while i < len(recs):
        n, j = 0, i
        while j < len(recs) and n + len(recs[j]) + 1 <= raw.size:
            n += len(recs[j]) + 1
            j += 1
        assert j > i
        raw[:n] = np.frombuffer(b"\n".join(recs[i:j]) + b"\n", dtype=np.uint8)
        if tickets[k] is not None:
            ctx.ticket_wait(tickets[k])
        _, odd = sk.pack_stream(raw[:n], out=pk[k])
        assert not odd
        tickets[k] = ctx.scan_pinned_packed(pk[k], n, 2)
        npacked += 1
        k ^= 1
        i = j
    ctx.sync()
    assert npacked >= 1
    assert np.array_equal(ctx.counts(1), ctx.counts(2)) and int(ctx.counts(1).sum()) > 5000
    t = _oracle.OracleTable(ncols=4)
    assert t.build_stream(strain + b"\n") == 0
    t.scan_stream(data, 1)
    okeys, ocounts = t.rows()
    assert okeys == ks.keys() and np.array_equal(ocounts[:, 1], ctx.counts(2))
    t.close()
    assert sk.pack_stream(b"ACGTRACGT\n")[1] and sk.pack_stream(b"ACGU\n")[1] and sk.pack_stream(b"ACGT\r\n")[1] and not sk.pack_stream(b"ACGTNn\n")[1]
    ctx.pinned_free(raw)
    for a in pk:
        ctx.pinned_free(a)


def test_rccl_allreduce_on_the_counter_block_single_rank(repo):
    """torch.distributed (backend nccl = RCCL) all-reduce running directly on the library's device counters,
    as bench.py --gpus N does: world of one, counts unchanged (tools/nccl_single_rank_check.py)"""
    import socket
    with socket.socket() as sock:                      # a port that is free right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "nccl_single_rank_check.py")], cwd=repo,
                       capture_output=True, timeout=300, env=dict(os.environ, MASTER_PORT=str(port)))
    assert p.returncode == 0 and b"all-reduce on the library's counter block: ok" in p.stdout, p.stderr.decode()[-800:]


def test_strain_with_more_keys_than_2_pow_24():
    """A 20 Mbp strain: 20 M keys, 2^26 table slots.  (The home slot once took 24 bits of the hash: every key of such a
    table started in the first 2^24 slots and the table load never finished.)  No oracle at this size -- the check is
    exact all the same: reads cut out of the strain give 120 hits each, one per window, each on its own row with the
    strain's own count; random reads give none."""
    import time
    rng = np.random.default_rng(2024)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    strain = acgt[rng.integers(0, 4, 20_000_000)]
    t = time.time()
    ks = sk.Keyset.from_stream(strain.tobytes() + b"\n")
    assert ks.nrows > (1 << 24)
    n = 20_000
    starts = rng.integers(0, len(strain) - 150, n)
    reads = strain[starts[:, None] + np.arange(150)[None, :]]
    flip = rng.random(n) < 0.5
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    reads[flip] = comp[reads[flip]][:, ::-1]
    noise = acgt[rng.integers(0, 4, (n, 150))]
    stream = b"".join(r.tobytes() + b"\n" for r in np.concatenate([reads, noise]))
    with sk.KmerContext(0) as c:
        c.load_keyset(ks, 4)
        assert time.time() - t < 120
        c.scan_stream(stream, 2)
        counts = c.counts(2)
    assert int(counts.sum()) == 120 * n
    # every window of read 0 (forward strand copy of the strain at starts[0]) is counted at least once
    keys = ks.keys()
    assert len(keys) == ks.nrows


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_rank_sharded_list_adds_up_on_the_device(world, tmp_path, monkeypatch):
    """SURVEY 8(e) from the product's side: every rank scans ITS share of a skewed list (skh_scan_list's own dealing
    by size and cutting of the big file), the ranks' raw counter blocks -- as they lie on the device, locality order,
    exactly what the RCCL all-reduce sums -- are added mod 2^32, and the result is the unsharded block and the oracle's
    counts.  (Several contexts on one device stand in for the ranks.)"""
    rng = random.Random(4242 + world)
    strain = _synth.rand_dna(rng, 80_000)
    sstream = strain + b"\n"
    sizes = [9000, 300, 1200, 40, 700]                                  # reads per file: one file dominates
    names = []
    for i, n in enumerate(sizes):
        data = _synth.fuzz_stream(rng, strain, n, p_junk=0.002, min_len=20, max_len=300)
        recs = [r for r in data.split(b"\n") if r]
        p = tmp_path / f"f{i}.fa"
        p.write_bytes(b"".join(b">r%d\n%s\n" % (j, r) for j, r in enumerate(recs)))
        names.append(str(p))
    lst = tmp_path / "list.txt"
    lst.write_text("".join(n + "\n" for n in names))
    monkeypatch.setenv("SK_SPLIT_BYTES", "200000")                      # so that the big file IS cut into pieces
    monkeypatch.setenv("SK_THREADS", "3")
    ks = sk.Keyset.from_stream(sstream)

    def block(rank, nranks):
        with sk.KmerContext(0) as c:
            c.load_keyset(ks, 4)
            c.zero_counts(0)                                            # (column 0 holds the build counts: not part of the sum)
            c.scan_list(str(lst), 2, rank=rank, world=nranks)
            c.sync()
            raw = c.dev_download(c.counts_device_ptr(), 4 * 4 * ks.nrows).view(np.uint32).copy()
            return raw, c.counts(2)

    whole, whole_rows = block(0, 1)
    total = np.zeros_like(whole)
    for r in range(world):
        total += block(r, world)[0]                                     # (uint32: wraps like the all-reduce)
    assert np.array_equal(total, whole)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    for n in names:
        for rec in open(n, "rb").read().split(b"\n"):
            if rec and not rec.startswith(b">"):
                t.scan_stream(rec + b"\n", 2)
    _, ocounts = t.rows()
    assert np.array_equal(whole_rows, ocounts[:, 2]) and whole_rows.sum() > 100_000


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "5", "16"])
def test_program_on_the_cfg3_shaped_job_equals_the_reference(golden, tmp_path, threads, monkeypatch):
    """BASELINE configs[2] in shape through bin/kmer_scrub_count: a real 1000-line -A list, a multi-file -B list (plain and
    .gz FASTQ), a -C list that holds the -r path (skip rule), -p.  stdout md5, stderr and the progress file (without its time
    stamps) must be what the UNMODIFIED reference program produced for the same inputs (tests/golden/cfg3_shape_facts.json;
    src/kmer_scrub_count.c:29-156, src/genome_compare.c:115-236) -- with one decode thread (the reference's strict sequence),
    a few, and sixteen."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cfg3_shape", os.path.join(golden, "make_cfg3_shape.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    facts = json.load(open(os.path.join(golden, "cfg3_shape_facts.json")))
    argv = mk.write_inputs(str(tmp_path))
    monkeypatch.setenv("SK_THREADS", threads)
    p = subprocess.run([sk.cli_path()] + argv, cwd=str(tmp_path), capture_output=True)
    assert p.returncode == facts["returncode"], p.stderr.decode()[-500:]
    assert p.stderr.decode() == facts["stderr"]
    got = mk.facts_of(p.stdout, p.stderr, str(tmp_path))
    for k in ("md5_stdout", "lines", "column_sums", "md5_progress_without_times"):
        assert got[k] == facts[k], k


@pytest.mark.gpu
def test_counters_wrap_and_print_as_signed_through_the_gpu_scan(tmp_path):
    """src/kmer_scrub_count.c:146-151 prints the unsigned counters with %d and they wrap mod 2^32: counters preset just
    below 2^31 and just below 2^32 are scanned over on the device (run-length updates folded in by a prefix sum, plus the
    one-by-one path) and printed by skh_print_counts; expected = (preset + oracle's increments) mod 2^32 as int32."""
    rng = random.Random(99)
    strain = _synth.rand_dna(rng, 40_000)
    sstream = strain + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    data = _synth.fuzz_stream(rng, strain, 4000, p_junk=0.002, min_len=25, max_len=260)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    t.scan_stream(data, 2)
    okeys, ocounts = t.rows()
    inc = ocounts[:, 2].astype(np.uint32)
    assert (inc > 2).sum() > 1000
    preset = np.zeros(ks.nrows, dtype=np.uint32)
    preset[0::3] = 0xFFFFFFFF                       # wraps to inc - 1
    preset[1::3] = 0x7FFFFFFE                       # crosses into the negative %d range
    preset[2::3] = 7
    with sk.KmerContext(0) as c:
        c.load_keyset(ks, 4)
        c.set_counts(2, preset)
        c.scan_stream(data, 2)
        got = c.counts(2)
        out = tmp_path / "table.tsv"
        c.print_counts(ks, str(out))
        ref_col, pan_col = c.counts(0), c.counts(1)
    want = (preset.astype(np.uint64) + inc.astype(np.uint64)).astype(np.uint32)       # mod 2^32
    assert np.array_equal(got, want)
    lines = open(out, "rb").read().split(b"\n")
    assert lines[0] == b"#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count"
    signed = want.view(np.int32)
    assert (signed < 0).sum() > 1000
    for r in range(0, ks.nrows, 97):
        assert lines[1 + r] == b"%s\t%d\t%d\t%d" % (okeys[r], int(ref_col[r]), int(pan_col[r]), int(signed[r])), r


@pytest.mark.gpu
@pytest.mark.parametrize("sub_rate", [0.01, 0.03, 0.10])
def test_diverged_copies_of_the_strain_vs_oracle(sub_rate):
    """every read a copy of a strain segment with 1 / 3 / 10 % substitutions, both strands, long and short: the regime of
    near-copy genomes in an -A list.  Stage 2 explains what it can by the diagonal, rejects the windows around a
    differing base by three filter questions per base, and sends the rest one by one -- the counts must be the oracle's."""
    rng = random.Random(int(sub_rate * 1000))
    strain = _synth.rand_dna(rng, 150_000)
    sstream = strain[:70_000] + b"\n" + strain[70_000:] + b"\n"
    recs = []
    for i in range(2500):
        L = rng.choice([40, 100, 150, 150, 400, 3000])
        a = rng.randrange(0, len(strain) - L)
        r = bytearray(strain[a:a + L])
        for j in range(L):
            if rng.random() < sub_rate:
                r[j] = rng.choice(b"ACGT")
        recs.append(_synth.revcomp(bytes(r)) if rng.random() < 0.5 else bytes(r))
    data = b"\n".join(recs) + b"\n"
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream) == 0
    t.scan_stream(data, 2)
    _, ocounts = t.rows()
    for text_stage in (1, 0):
        with sk.KmerContext(0) as c:
            c.set_option("text_stage", text_stage)
            c.load_keyset(ks, 4)
            c.scan_stream(data, 2)
            got = c.counts(2)
        assert np.array_equal(got, ocounts[:, 2]), text_stage
    assert ocounts[:, 2].sum() > 20_000
