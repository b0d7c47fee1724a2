"""ASan + UBSan runs on the CPU: the oracle over the golden cases, and the GPU-free parts of the host
layer (parser, keyset build, order replay, stream writer) over every fixture file."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")


def _fixture_files(golden):
    out = []
    for sub in ("cases", "sd_cases"):
        for root, _d, files in os.walk(os.path.join(golden, sub)):
            out += [os.path.join(root, f) for f in files if f.endswith((".fa", ".fasta", ".fq", ".gz", ".fx"))]
    return sorted(out)


def test_host_layer_under_asan_ubsan(golden, tmp_path):
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["gcc"] + SAN + [os.path.join(REPO, "tests", "native", "host_sanitize.c"),
                                   os.path.join(REPO, "strainer2_amd", "csrc", "sk_host.c"), "-lz", "-lpthread", "-o", exe], check=True)
    files = _fixture_files(golden)
    assert len(files) > 30
    p = subprocess.run([exe] + files, env=ENV, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr


@pytest.mark.parametrize("name", ["mixed", "drug", "iupac_strain", "truncated_fastq", "contig30"])
def test_oracle_under_asan_ubsan(golden, name, tmp_path):
    exe = str(tmp_path / "kso_asan")
    subprocess.run(["gcc"] + SAN + ["-DKSO_MAIN", os.path.join(REPO, "oracle", "kso_oracle.c"), "-lz", "-o", exe], check=True)
    d = os.path.join(golden, "cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = [a if a != "progress.txt" and a != "prog.txt" else str(tmp_path / "p") for a in meta["argv"]]
    p = subprocess.run([exe] + argv, cwd=d, env=ENV, capture_output=True)
    assert p.returncode == meta["returncode"], p.stderr.decode()[-2000:]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()


@pytest.fixture(scope="module")
def filter_host_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("flt") / "filter_sanitize")
    subprocess.run(["gcc"] + SAN + [os.path.join(REPO, "tests", "native", "filter_sanitize.c"),
                                   os.path.join(REPO, "strainer2_amd", "csrc", "sk_host_filter.c"), "-lz", "-o", exe], check=True)
    return exe


FILTER_CASES = os.path.join(REPO, "tests", "golden", "filter_cases")


@pytest.mark.parametrize("name", sorted(os.listdir(FILTER_CASES)))
def test_filter_host_logic_under_asan_ubsan(filter_host_exe, name):
    """Host half of kmer_scrub_filter (parse, dictionaries, the float cut, printing) against the reference
    script's output, with a test double for the device calls (tests/native/filter_sanitize.c)."""
    d = os.path.join(FILTER_CASES, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    p = subprocess.run([filter_host_exe] + meta["argv"], cwd=d, env=ENV, capture_output=True)
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
    assert p.returncode == meta["returncode"], p.stderr.decode()[-2000:]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    if meta["stderr_exact"]:
        assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
    elif meta["returncode"]:
        assert p.stderr


@pytest.fixture(scope="module")
def cov_host_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cov") / "cov_sanitize")
    subprocess.run(["gcc"] + SAN + [os.path.join(REPO, "tests", "native", "cov_sanitize.c"),
                                   os.path.join(REPO, "strainer2_amd", "csrc", "sk_host_cov.c"), "-lz", "-o", exe], check=True)
    return exe


COV_CASES = os.path.join(REPO, "tests", "golden", "cov_cases")


def prepare_cov_case(name, d):
    """the bundled step-4 case reads the step-3 golden, kept once under tests/golden/bundled"""
    if name == "bundled_step4":
        import gzip
        dst = os.path.join(d, "Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208.kmer_hits.gz")
        if not os.path.exists(dst):
            with open(os.path.join(REPO, "tests", "golden", "bundled", "step3_expected.hits"), "rb") as f, \
                    gzip.GzipFile(dst, "wb", mtime=0) as g:
                g.write(f.read())


@pytest.mark.parametrize("name", sorted(os.listdir(COV_CASES)))
def test_coverage_host_logic_under_asan_ubsan(cov_host_exe, name):
    d = os.path.join(COV_CASES, name)
    prepare_cov_case(name, d)
    meta = json.load(open(os.path.join(d, "case.json")))
    p = subprocess.run([cov_host_exe] + meta["argv"], cwd=d, env=ENV, capture_output=True)
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
    assert p.returncode == meta["returncode"], p.stderr.decode()[-2000:]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()


SD_DIR = os.path.join(REPO, "tests", "golden", "sd_cases")
SD_SOURCES = [os.path.join(REPO, "tests", "native", "device_double.c")] + \
    [os.path.join(REPO, "strainer2_amd", "csrc", f) for f in ("sk_host.c", "sk_host_sd.c", "sk_host_cov.c")]


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def sd_host_exe(request, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sd") / ("sd_" + request.param.split(",")[0]))
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=" + request.param, "-fno-omit-frame-pointer"] + SD_SOURCES +
                   ["-lz", "-lpthread", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("name", ["batch", "cli_se", "cli_pe", "cli_pei", "cli_default", "err_missing", "err_type",
                                  "err_pe_one_file", "err_b_and_B", "err_no_inf", "err_no_read1"])
def test_strain_detect_host_logic_under_sanitizers(sd_host_exe, name, tmp_path):
    """reader threads, chunk queue, replay, thread pool and the fused coverage table of strain_detect under
    ASan+UBSan and under TSan, with a CPU test double for the device calls; outputs against the reference's."""
    import gzip
    d = os.path.join(SD_DIR, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = list(meta["argv"])
    if "-o" in argv:
        argv[argv.index("-o") + 1] = str(tmp_path / "a_b_c.kmer_hits.gz")
        argv += ["--coverage-depth"]
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4")
    p = subprocess.run([sd_host_exe] + argv, cwd=d, env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert p.returncode == meta["returncode"], p.stderr.decode()[-2000:]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
    if meta["returncode"] == 0:
        assert gzip.open(tmp_path / "a_b_c.kmer_hits.gz", "rb").read() == open(os.path.join(d, "expected.hits"), "rb").read()
        assert open(tmp_path / "a_b_c.coverage_depth", "rb").read().startswith(b"strain_name\t")


@pytest.mark.parametrize("name", ["batch", "cli_se", "cli_pe", "cli_pei"])
@pytest.mark.parametrize("chunk", ["64", "333", "2000"])
def test_strain_detect_tiny_chunks_under_sanitizers(sd_host_exe, name, chunk, tmp_path):
    """chunks of a few reads: mates that sit in different chunks (PE and interleaved), tallies and the PE1 copy
    carried across chunk boundaries, queues that fill up -- same output as with one big chunk"""
    import gzip
    d = os.path.join(SD_DIR, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = list(meta["argv"])
    argv[argv.index("-o") + 1] = str(tmp_path / "o.gz")
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_SD_CHUNK_BYTES=chunk)
    p = subprocess.run([sd_host_exe] + argv, cwd=d, env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert p.returncode == meta["returncode"] == 0
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    assert gzip.open(tmp_path / "o.gz", "rb").read() == open(os.path.join(d, "expected.hits"), "rb").read()


@pytest.mark.parametrize("union", [True, False])
@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_strain_detect_many_strains_host_logic_under_sanitizers(san, union, tmp_path):
    """-S with four strains (the thread pool is in use): every output equals the single-strain golden; the results
    of a batch dealt out from a union table (the double merges its members' answers into the union's format), and
    fetched strain by strain"""
    import gzip
    exe = str(tmp_path / "sd_multi")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer"] + SD_SOURCES + ["-lz", "-lpthread", "-o", exe], check=True)
    d = os.path.join(SD_DIR, "batch")
    with open(tmp_path / "strains.txt", "w") as f:
        for s in range(4):
            f.write(f"strain.fa\tinf.txt.gz\t{tmp_path}/o{s}.gz\n")
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_SD_TIMING="1")
    if not union:
        env["DOUBLE_NO_UNION"] = "1"
    p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", "B.txt", "--coverage-depth"], cwd=d, env=env, capture_output=True)
    assert (b"1 union table(s) for 4 strains" in p.stderr) == union
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert p.returncode == 0
    want = open(os.path.join(d, "expected.hits"), "rb").read()
    for s in range(4):
        assert gzip.open(tmp_path / f"o{s}.gz", "rb").read() == want


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_strain_detect_one_decoder_several_devices_under_sanitizers(san, tmp_path):
    """SK_DEVICES: ONE process and ONE decode pipeline drive several devices -- the strains are dealt to the devices in groups
    (one union table per group; SK_SD_GROUP=2 makes groups of two so that five strains are three groups on three logical
    devices), every decoded chunk goes up to every device, each device tallies it against its own strains.  Every strain's
    file must be the single-strain golden.  (CPU double for the device calls: the host logic is what runs here -- per-device
    batches, the prefetch of the next chunk to all devices, results dealt back to the strains.)"""
    import gzip
    exe = str(tmp_path / "sd_multi")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer"] + SD_SOURCES + ["-lz", "-lpthread", "-o", exe], check=True)
    d = os.path.join(SD_DIR, "batch")
    with open(tmp_path / "strains.txt", "w") as f:
        for s in range(5):
            f.write(f"strain.fa\tinf.txt.gz\t{tmp_path}/o{s}.gz\n")
    want = open(os.path.join(d, "expected.hits"), "rb").read()
    for devices, chunk in (("3", "700"), ("0,0", "4000")):
        env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_SD_TIMING="1", SK_DEVICES=devices, SK_SD_GROUP="2", SK_SD_CHUNK_BYTES=chunk)
        p = subprocess.run([exe, "-S", str(tmp_path / "strains.txt"), "-B", "B.txt"], cwd=d, env=env, capture_output=True)
        for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
            assert bad not in p.stderr, p.stderr.decode()[-3000:]
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"3 union table(s) for 5 strains" in p.stderr
        assert (b"5 strains on %d devices, one decode pipeline" % (3 if devices == "3" else 2)) in p.stderr
        for s in range(5):
            assert gzip.open(tmp_path / f"o{s}.gz", "rb").read() == want
            os.remove(tmp_path / f"o{s}.gz")


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def ks_host_exe(request, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ks") / ("ks_" + request.param.split(",")[0]))
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=" + request.param, "-fno-omit-frame-pointer", "-DDOUBLE_MAIN=skh_kmer_scrub_count_main"] +
                   SD_SOURCES + ["-lz", "-lpthread", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("name", ["truncated_fastq", "contig30", "missing_in_list", "missing_flag", "progress_missing", "skip_after_missing"])
def test_kmer_scrub_count_host_logic_under_sanitizers(ks_host_exe, golden, name, tmp_path):
    """kmer_scrub_count's host half (key-set build, order replay, decode thread pool with its double buffers,
    table print) under ASan+UBSan and TSan with the CPU test double, against the reference's output"""
    d = os.path.join(golden, "cases", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_CHUNK_BYTES="4096")     # (many chunk flushes per file, both buffers in use)
    argv = [a if a != "progress.txt" else str(tmp_path / "progress") for a in meta["argv"]]
    p = subprocess.run([ks_host_exe] + argv, cwd=d, env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert p.returncode == meta["returncode"]
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    assert p.stderr == open(os.path.join(d, "expected.stderr"), "rb").read()
    if meta["progress_col1"] is not None:
        # with four decode threads the files behind the unreadable one are taken (and announced) meanwhile: the progress
        # file must still end where the reference's does (src/genome_compare.c:167-172,195-198)
        with open(tmp_path / "progress") as f:
            assert [ln.split("\t")[0].rstrip("\n") for ln in f] == meta["progress_col1"]


def test_kmer_scrub_count_thread_pool_under_sanitizers(ks_host_exe, tmp_path):
    """24 list files over 4 decode threads: same table as the oracle program"""
    import random
    rng = random.Random(5)
    strain = "".join(rng.choice("ACGT") for _ in range(20000))
    (tmp_path / "s.fa").write_text(">s\n" + strain + "\n")
    names = []
    for i in range(24):
        recs = []
        for j in range(rng.randrange(1, 400)):
            a = rng.randrange(0, len(strain) - 200)
            recs.append(f">r{j}\n{strain[a:a + rng.randrange(20, 200)]}\n")
        (tmp_path / f"m{i}.fa").write_text("".join(recs))
        names.append(f"m{i}.fa")
    (tmp_path / "A.txt").write_text("\n".join(names[:5]) + "\n")
    (tmp_path / "B.txt").write_text("\n".join(names) + "\n")
    argv = ["-r", "s.fa", "-A", "A.txt", "-B", "B.txt"]
    oracle = os.path.join(REPO, "oracle", "kso_oracle")
    if not os.path.exists(oracle):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "kso_oracle"], check=True, stdout=subprocess.DEVNULL)
    want = subprocess.run([oracle] + argv, cwd=tmp_path, capture_output=True)
    for chunk in ("4096", "33554432"):                       # chunks of a few records (many flushes, both buffers in turn) and of a whole file
        env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_CHUNK_BYTES=chunk)
        p = subprocess.run([ks_host_exe] + argv, cwd=tmp_path, env=env, capture_output=True)
        for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
            assert bad not in p.stderr, p.stderr.decode()[-3000:]
        assert (p.returncode, p.stdout) == (want.returncode, want.stdout) and want.returncode == 0


@pytest.mark.parametrize("name", ["batch", "cli_pe", "cli_pei"])
def test_strain_detect_members_inflated_by_several_threads(sd_host_exe, name, tmp_path):
    """the read files of strain_detect through sk_gzpar.h (three inflating threads per file, segments of 100 bytes
    so that the small fixtures are cut into many): same output"""
    import gzip
    d = os.path.join(SD_DIR, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = list(meta["argv"])
    argv[argv.index("-o") + 1] = str(tmp_path / "o.gz")
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_GZ_THREADS="3", SK_GZ_SEG="100")
    p = subprocess.run([sd_host_exe] + argv, cwd=d, env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert p.returncode == meta["returncode"] == 0
    assert p.stdout == open(os.path.join(d, "expected.stdout"), "rb").read()
    assert gzip.open(tmp_path / "o.gz", "rb").read() == open(os.path.join(d, "expected.hits"), "rb").read()


def test_kmer_scrub_count_members_inflated_by_several_threads(ks_host_exe, tmp_path):
    """three .gz list files, each inflated by three threads in segments of 200 bytes, one of them cut short: the
    oracle program's table and messages"""
    import gzip
    import random
    rng = random.Random(6)
    strain = "".join(rng.choice("ACGT") for _ in range(20000))
    (tmp_path / "s.fa").write_text(">s\n" + strain + "\n")
    names = []
    for i in range(3):
        recs = []
        for j in range(600):
            a = rng.randrange(0, len(strain) - 200)
            seq = strain[a:a + rng.randrange(20, 200)]
            recs.append(f"@r{j}\n{seq}\n+\n{'I' * len(seq)}\n")
        blob = gzip.compress("".join(recs).encode(), 6, mtime=0)
        (tmp_path / f"m{i}.fq.gz").write_bytes(blob if i < 2 else blob[:len(blob) * 2 // 3])
        names.append(f"m{i}.fq.gz")
    (tmp_path / "A.txt").write_text(names[0] + "\n")
    (tmp_path / "B.txt").write_text("\n".join(names) + "\n")
    argv = ["-r", "s.fa", "-A", "A.txt", "-B", "B.txt"]
    oracle = os.path.join(REPO, "oracle", "kso_oracle")
    if not os.path.exists(oracle):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "kso_oracle"], check=True, stdout=subprocess.DEVNULL)
    want = subprocess.run([oracle] + argv, cwd=tmp_path, capture_output=True)
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_GZ_THREADS="3", SK_GZ_SEG="200")
    p = subprocess.run([ks_host_exe] + argv, cwd=tmp_path, env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in p.stderr, p.stderr.decode()[-3000:]
    assert (p.returncode, p.stdout, p.stderr) == (want.returncode, want.stdout, want.stderr)


def _sd_inputs(tmp_path, seed):
    """a strain with an informative list and three read files whose records test the cutting: FASTQ with quality lines that
    begin with @ + >, wrapped FASTA, short reads (the carried-over tallies of the reference), an odd interleaved file"""
    import gzip
    import random
    import _synth
    rng = random.Random(seed)
    strain = _synth.rand_dna(rng, 30_000)
    (tmp_path / "s.fa").write_bytes(b">s\n" + strain + b"\n")

    def canon(k):
        r = _synth.revcomp(k)
        return k if k >= r else r
    kms = sorted({canon(strain[i:i + 31]) for i in range(0, len(strain) - 31, 13)})
    with gzip.open(tmp_path / "s.inf.gz", "wb") as f:
        f.write(b"\n".join(kms) + b"\n")

    def read(r):
        L = r.choice([20, 31, 75, 150, 150, 151, 300])
        if r.random() < 0.4:
            a = r.randrange(0, len(strain) - L)
            s = strain[a:a + L]
            return _synth.revcomp(s) if r.random() < 0.5 else s
        return _synth.rand_dna(r, L)

    def fastq(n, r):
        out = []
        for i in range(n):
            s = read(r)
            out.append(b"@q%d x\n%s\n+\n%s\n" % (i, s, bytes(r.choice(b"@+>IF#") for _ in s)))
        return b"".join(out)

    def fasta(n, r, width):
        return b"".join(b">f%d\n" % i + (b"\n".join(s[j:j + width] for j in range(0, len(s), width)) if width else s) + b"\n"
                        for i, s in ((i, read(r)) for i in range(n)))
    with gzip.open(tmp_path / "se.fq.gz", "wb", compresslevel=1) as f:
        f.write(fastq(2500, random.Random(seed + 1)))
    (tmp_path / "pe_1.fq").write_bytes(fastq(1200, random.Random(seed + 2)))
    (tmp_path / "pe_2.fa").write_bytes(fasta(1200, random.Random(seed + 3), 60))
    (tmp_path / "il.fa").write_bytes(fasta(1501, random.Random(seed + 4), 0))
    (tmp_path / "B.txt").write_text(f"SE\t{tmp_path}/se.fq.gz\nPE\t{tmp_path}/pe_1.fq\t{tmp_path}/pe_2.fa\nPEI\t{tmp_path}/il.fa\n")


@pytest.mark.parametrize("chunk", ["700", "5000", "60000"])
def test_strain_detect_parser_threads_give_the_serial_result(sd_host_exe, tmp_path, chunk):
    """several parser threads on one metagenome file (segments cut at guessed record starts, each CHECKED to end between two
    records, chunks queued in file order; sk_host_sd.c) against the one-thread decode of the same program and against the
    CPU oracle program -- under ASan+UBSan and TSan, with segments of a few hundred bytes to tens of kilobytes"""
    import gzip
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import _oracle
    _sd_inputs(tmp_path, 77)
    base = ["-r", str(tmp_path / "s.fa"), "-a", str(tmp_path / "s.inf.gz"), "-B", str(tmp_path / "B.txt")]
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_SD_CHUNK_BYTES=chunk, SK_PARSE_THREADS="3", SK_READ_BLOCK="257")      # (257: a segment is pread in many pieces, cut anywhere)
    env["SK_SD_INPUT"] = {"700": "pread", "5000": "mapped"}.get(chunk, "populate")      # the three ways a parser thread gets at a mapped file's segment
    par = subprocess.run([sd_host_exe] + base + ["-o", str(tmp_path / "par.gz")], env=env, capture_output=True)
    for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert bad not in par.stderr, par.stderr.decode()[-3000:]
    assert par.returncode == 0, par.stderr.decode()[-500:]
    ser = subprocess.run([sd_host_exe] + base + ["-o", str(tmp_path / "ser.gz")], env=dict(env, SK_NO_SPLIT="1"), capture_output=True)
    assert ser.returncode == 0 and ser.stdout == par.stdout
    a, b = gzip.open(tmp_path / "par.gz", "rb").read(), gzip.open(tmp_path / "ser.gz", "rb").read()
    assert a == b and a.count(b"\n") > 500
    ora = _oracle.run_sd_oracle_cli(base + ["-o", str(tmp_path / "ora.gz")], str(tmp_path))
    assert ora.returncode == 0 and gzip.open(tmp_path / "ora.gz", "rb").read() == a


def test_kmer_scrub_count_one_gz_parsed_by_several_threads(ks_host_exe, tmp_path):
    """one .gz list entry inflated by three threads (SK_GZ_THREADS) whose text is cut into segments at guessed record starts
    and parsed by helper threads (parse_gz_split, sk_host.c): FASTQ with quality lines that begin with @ + >, wrapped FASTA,
    CRLF -- same table as the oracle program; a FASTQ record that ends the file for the reference in the middle of the
    file fails the cut scan and goes through uncut (SK_NO_SPLIT=1) with the oracle's table"""
    import gzip
    import random
    rng = random.Random(77)
    strain = "".join(rng.choice("ACGT") for _ in range(30000))
    (tmp_path / "s.fa").write_text(">s\n" + strain + "\n")

    def fastq(n, seed):
        r = random.Random(seed)
        out = []
        for i in range(n):
            L = r.choice([31, 60, 150, 150, 250])
            a = r.randrange(0, len(strain) - L)
            q = "".join(r.choice("@+>IIIIFF#") for _ in range(L))
            out.append(f"@r{i} c\n{strain[a:a + L]}\n+\n{q}\n")
        return "".join(out)

    def fasta(n, seed, width):
        r = random.Random(seed)
        out = []
        for i in range(n):
            L = r.randrange(100, 4000)
            a = r.randrange(0, len(strain) - L)
            seq = strain[a:a + L]
            out.append(f">c{i}\n" + "\n".join(seq[j:j + width] for j in range(0, L, width)) + "\n")
        return "".join(out)

    files = {"a.fq.gz": fastq(1500, 1), "b.fa.gz": fasta(120, 2, 60), "c.fq.gz": fastq(400, 3).replace("\n", "\r\n")}
    for name, text in files.items():
        with gzip.open(tmp_path / name, "wb", compresslevel=6) as f:
            f.write(text.encode())
    (tmp_path / "A.txt").write_text("b.fa.gz\n")
    (tmp_path / "B.txt").write_text("a.fq.gz\nc.fq.gz\n")
    argv = ["-r", "s.fa", "-A", "A.txt", "-B", "B.txt"]
    oracle = os.path.join(REPO, "oracle", "kso_oracle")
    if not os.path.exists(oracle):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "kso_oracle"], check=True, stdout=subprocess.DEVNULL)
    want = subprocess.run([oracle] + argv, cwd=tmp_path, capture_output=True)
    assert want.returncode == 0
    for chunk, par in (("4096", "3"), ("20000", "8"), ("33554432", "2")):
        env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_GZ_THREADS="3", SK_GZ_SEG="3000", SK_CHUNK_BYTES=chunk, SK_PARSE_THREADS=par)
        p = subprocess.run([ks_host_exe] + argv, cwd=tmp_path, env=env, capture_output=True)
        for bad in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
            assert bad not in p.stderr, p.stderr.decode()[-3000:]
        assert (p.returncode, p.stdout) == (want.returncode, want.stdout), p.stderr.decode()[-1000:]
    # a record whose quality is longer than its sequence, in the middle
    text = fastq(600, 5) + "@bad\n" + strain[100:250] + "\n+\n" + "I" * 170 + "\n" + fastq(600, 6)
    with gzip.open(tmp_path / "d.fq.gz", "wb") as f:
        f.write(text.encode())
    (tmp_path / "B.txt").write_text("d.fq.gz\n")
    want = subprocess.run([oracle] + argv, cwd=tmp_path, capture_output=True)
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1", SK_THREADS="4", SK_GZ_THREADS="3", SK_GZ_SEG="3000", SK_CHUNK_BYTES="8192", SK_PARSE_THREADS="4")
    p = subprocess.run([ks_host_exe] + argv, cwd=tmp_path, env=env, capture_output=True)
    assert p.returncode != 0 and b"could not be cut at record boundaries" in p.stderr
    p = subprocess.run([ks_host_exe] + argv, cwd=tmp_path, env=dict(env, SK_NO_SPLIT="1"), capture_output=True)
    assert (p.returncode, p.stdout) == (want.returncode, want.stdout) and want.returncode == 0
