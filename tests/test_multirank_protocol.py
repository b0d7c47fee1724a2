"""The multi-rank kmer_scrub_count program as SEVERAL PROCESSES on the CPU (ADVICE r03 high + medium).

The real skh_kmer_scrub_count_main runs over the plain-C device double, whose communicator (tests/native/device_double.c,
DOUBLE_COMM_DIR) passes every collective between the processes through files, CHECKS that all ranks call the same
collective with the same element count in the same order (what RCCL needs to be true and cannot check: exit 98 on a
mismatch, exit 97 on a rank that never arrives) and logs each rank's sequence, which is compared line by line here.

What must hold (src/genome_compare.c:149-177,195-203 is one process; the sharding is new):
  * a sharded run prints the table of the one-process run, and every rank issues the same collectives;
  * a file only ONE rank owns is missing / a list only one rank cannot read / ranks with different SK_SPLIT_BYTES:
    every rank leaves with status 1 -- nobody waits in a collective the others never enter;
  * a big text file whose cut does not hold (inputs the reference accepts with exit 0, src/genome_compare.c:203) no longer
    fails a multi-rank run: all ranks put their columns back and scan the whole-file plan (round 3: SK_E_SPLIT).
Built with ASan + UBSan."""
import os
import random
import subprocess

import pytest

import _synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(REPO, "tests", "native", "device_double.c")] + \
      [os.path.join(REPO, "strainer2_amd", "csrc", f) for f in ("sk_host.c", "sk_host_sd.c", "sk_host_cov.c")]


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mr") / "ksc_double")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-DDOUBLE_MAIN=skh_kmer_scrub_count_main"] + SRC + ["-lz", "-lpthread", "-o", out], check=True)
    return out


def _fastq(rng, strain, n):
    out = []
    for i in range(n):
        L = rng.choice([31, 60, 150, 150, 150, 250])
        a = rng.randrange(0, len(strain) - L)
        s = strain[a:a + L] if rng.random() < 0.5 else _synth.rand_dna(rng, L)
        out.append(b"@r%d c\n%s\n+\n%s\n" % (i, s, b"I" * L))
    return b"".join(out)


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("mr_data")
    rng = random.Random(404)
    strain = _synth.rand_dna(rng, 40000)
    (d / "strain.fa").write_bytes(b">s\n" + strain + b"\n")
    for i in range(6):
        (d / f"m{i}.fq").write_bytes(_fastq(rng, strain, 200 + 300 * i))
    (d / "big.fq").write_bytes(_fastq(rng, strain, 5000))
    for i in range(3):
        a = rng.randrange(0, 20000)
        (d / f"g{i}.fa").write_bytes(b">g\n" + strain[a:a + 15000] + b"\n")
    # the two files of tests/test_sharding.py whose cuts do not hold
    recs = []
    for i in range(400):
        a = rng.randrange(0, len(strain) - 150)
        s = strain[a:a + 150]
        q = b"@" + b"I" * 49 + b"\n" + b"I" * 50 + b"\n" + b"+" + b"I" * 49          # quality lines that imitate a header
        recs.append(b"@r%d\n%s\n%s\n%s\n+\n%s\n" % (i, s[:50], s[50:100], s[100:], q))
    (d / "wrapped.fq").write_bytes(b"".join(recs))
    bad = b"@bad\n" + strain[100:250] + b"\n+\n" + b"I" * 170 + b"\n"                 # ends the file for the reference (kseq -2)
    (d / "whole.fq").write_bytes(_fastq(rng, strain, 300) + bad + _fastq(rng, strain, 300))
    (d / "A.txt").write_text("".join(str(d / f"g{i}.fa") + "\n" for i in range(3)))
    (d / "B.txt").write_text("".join(str(d / n) + "\n" for n in ["m0.fq", "big.fq", "m1.fq", "m2.fq", "m3.fq", "m4.fq", "m5.fq"]))
    (d / "C.txt").write_text(str(d / "g1.fa") + "\n" + str(d / "strain.fa") + "\n" + str(d / "g2.fa") + "\n")
    (d / "B_missing.txt").write_text("".join(str(d / n) + "\n" for n in ["m0.fq", "big.fq", "m1.fq", "nope.fq", "m2.fq", "m3.fq"]))
    (d / "B_adversarial.txt").write_text("".join(str(d / n) + "\n" for n in ["m0.fq", "wrapped.fq", "whole.fq", "m1.fq"]))
    return d


def run_ranks(exe, world, argv_of, env_of, tmp, timeout=120):
    """start the ranks together; returns [(returncode, stdout, stderr)], the ranks' collective logs"""
    comm = tmp / ("comm_%d" % len(list(tmp.iterdir())))
    comm.mkdir()
    ps = []
    for r in range(world):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", SK_THREADS="3", WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r),
                   DOUBLE_COMM_DIR=str(comm), DOUBLE_COMM_TIMEOUT="30")
        env.update(env_of(r))
        ps.append(subprocess.Popen([exe] + argv_of(r), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    out = []
    for p in ps:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise AssertionError("a rank hung")
        out.append((p.returncode, o, e))
    logs = [(comm / f"log.r{r}").read_text().splitlines() if (comm / f"log.r{r}").exists() else [] for r in range(world)]
    for rc, _o, e in out:
        assert rc not in (97, 98), e.decode()[-1500:]                   # (the double saw the ranks' sequences differ)
        assert b"AddressSanitizer" not in e and b"runtime error" not in e, e.decode()[-3000:]
    return out, logs


def single(exe, argv, **env):
    p = subprocess.run([exe] + argv, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", SK_THREADS="3", **env), capture_output=True, timeout=120)
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_program_prints_the_one_process_table(exe, data, tmp_path, world):
    argv = ["-r", str(data / "strain.fa"), "-A", str(data / "A.txt"), "-B", str(data / "B.txt"), "-C", str(data / "C.txt")]
    one = single(exe, argv + ["-p", str(tmp_path / "p1")], SK_NO_SPLIT="1")
    assert one.returncode == 0 and one.stdout.count(b"\n") > 30000
    out, logs = run_ranks(exe, world, lambda r: argv + (["-p", str(tmp_path / "pw")] if r == 0 else []),
                          lambda r: {"SK_SPLIT_BYTES": "60000"}, tmp_path)
    assert [rc for rc, _o, _e in out] == [0] * world, out[0][2][-1500:]
    assert out[0][1] == one.stdout and all(o == b"" for _rc, o, _e in out[1:])              # rank 0 alone prints
    assert out[0][2] == one.stderr and b"skipping" in one.stderr                           # ... and says what the reference says
    assert all(lg == logs[0] for lg in logs)
    # rendezvous, the table-load agreement, 3 lists x (before, after), the failure sum, the all-reduce
    assert [ln.split()[0] for ln in logs[0]] == ["rendezvous", "sum_u32"] + ["max_u64"] * 6 + ["sum_u32", "allreduce_u32"]
    col1 = lambda path: [ln.split("\t")[0].rstrip("\n") for ln in open(path)]              # noqa: E731
    assert col1(tmp_path / "pw") == col1(tmp_path / "p1")


def test_a_file_only_one_rank_owns_is_missing(exe, data, tmp_path):
    """the owner of nope.fq fails alone (SK_E_OPEN); before round 4 it skipped the -C list's agreement and went to
    the failure sum while the other rank sat in the agreement: mismatched all-reduces, a hang on the GPUs.  Now every rank
    learns of it in the agreement that closes the list, skips -C together and leaves with 1; the owner names the file once;
    rank 0's progress file ends with the unreadable file's line, as the reference's does (src/genome_compare.c:167-172,195-198)."""
    argv = ["-r", str(data / "strain.fa"), "-A", str(data / "A.txt"), "-B", str(data / "B_missing.txt"), "-C", str(data / "C.txt")]
    out, logs = run_ranks(exe, 2, lambda r: argv + (["-p", str(tmp_path / "pw")] if r == 0 else []), lambda r: {"SK_NO_SPLIT": "1"}, tmp_path)
    assert [rc for rc, _o, _e in out] == [1, 1]
    assert all(o == b"" for _rc, o, _e in out)
    said = b"".join(e for _rc, _o, e in out)
    assert said.count(b"could not read file %s in GEN_calculate_kmer_count()\n" % str(data / "nope.fq").encode()) == 1
    assert b"skipping" not in said                                                          # -C was never walked
    assert logs[0] == logs[1]
    assert [ln.split()[0] for ln in logs[0]] == ["rendezvous", "sum_u32"] + ["max_u64"] * 4 + ["sum_u32"]     # no -C agreements, no all-reduce
    lines = [ln.split("\t")[0].rstrip("\n") for ln in open(tmp_path / "pw")]
    assert lines[-1] == str(data / "nope.fq") and str(data / "m2.fq") not in lines


def test_a_list_only_one_rank_cannot_read(exe, data, tmp_path):
    argv = lambda r: ["-r", str(data / "strain.fa"), "-A", str(data / "A.txt"), "-B", str(data / ("B.txt" if r == 0 else "no_such_list.txt"))]   # noqa: E731
    out, logs = run_ranks(exe, 2, argv, lambda r: {"SK_NO_SPLIT": "1"}, tmp_path)
    assert [rc for rc, _o, _e in out] == [1, 1] and all(o == b"" for _rc, o, _e in out)
    assert b"could not read file %s in GEN_all_kmer_counts()" % str(data / "no_such_list.txt").encode() in out[1][2]
    assert b"another rank could not read" in out[0][2]
    assert logs[0] == logs[1]


def test_ranks_with_different_plans_all_leave(exe, data, tmp_path):
    """the SK_E_PLAN branch, exercised with values that really travel between the ranks"""
    argv = ["-r", str(data / "strain.fa"), "-A", str(data / "A.txt"), "-B", str(data / "B.txt")]
    out, logs = run_ranks(exe, 2, lambda r: argv, lambda r: {"SK_SPLIT_BYTES": "60000" if r == 0 else "90000"}, tmp_path)
    assert [rc for rc, _o, _e in out] == [1, 1] and all(o == b"" for _rc, o, _e in out)
    assert all(b"the ranks computed different work plans" in e for _rc, _o, e in out)
    assert logs[0] == logs[1]
    # -A (no pieces: same plan) goes through, -B's first agreement stops everybody
    assert [ln.split()[0] for ln in logs[0]] == ["rendezvous", "sum_u32", "max_u64", "max_u64", "max_u64", "sum_u32"]


def test_a_rank_whose_strain_is_unreadable(exe, data, tmp_path):
    argv = lambda r: ["-r", str(data / ("strain.fa" if r == 1 else "no_strain.fa")), "-A", str(data / "A.txt"), "-B", str(data / "B.txt")]   # noqa: E731
    out, logs = run_ranks(exe, 2, argv, lambda r: {}, tmp_path)
    assert [rc for rc, _o, _e in out] == [1, 1] and all(o == b"" for _rc, o, _e in out)
    assert logs[0] == logs[1] == ["rendezvous 1"]


@pytest.mark.parametrize("world", [2, 3])
def test_a_cut_that_does_not_hold_no_longer_fails_a_multi_rank_run(exe, data, tmp_path, world):
    """with world > 1 a byte-range cut that failed its check failed the whole run (SK_E_SPLIT) on inputs the
    reference accepts with exit 0 (src/genome_compare.c:203: kseq's -2 ends the file silently).  Now: every rank copies its
    column when the plan has pieces, all learn of the failed check in the closing agreement, all put the column back and scan
    their share of the whole-file plan -- the one-process table, byte for byte, and nothing on stderr."""
    argv = ["-r", str(data / "strain.fa"), "-A", str(data / "A.txt"), "-B", str(data / "B_adversarial.txt")]
    one = single(exe, argv, SK_NO_SPLIT="1")
    assert one.returncode == 0
    out, logs = run_ranks(exe, world, lambda r: argv, lambda r: {"SK_SPLIT_BYTES": "5000", "SK_TIMING": "1"}, tmp_path)
    assert [rc for rc, _o, _e in out] == [0] * world, out[0][2][-1500:]
    assert out[0][1] == one.stdout
    assert all(b"did not hold; the list is scanned again uncut" in e for _rc, _o, e in out)      # every rank went round again
    assert not any(b"could not be cut" in e for _rc, _o, e in out)
    assert all(lg == logs[0] for lg in logs)
    # -A: 2 agreements (it may have pieces at this size too: then 4); -B: before, after (cut failed), before, after
    kinds = [ln.split()[0] for ln in logs[0]]
    assert kinds[:2] == ["rendezvous", "sum_u32"] and kinds[-2:] == ["sum_u32", "allreduce_u32"] and kinds.count("max_u64") >= 6
