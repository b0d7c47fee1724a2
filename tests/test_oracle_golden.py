"""The oracle against the golden vectors (all produced by the unmodified reference binary,
see tests/golden/make_golden.py) and, where oracle/_ref is present, against that binary live."""
import hashlib
import json
import os
import random

import pytest

import _oracle
import _synth

CASES = ["mixed", "drug", "iupac_strain", "truncated_fastq", "missing_in_list", "missing_flag",
         "short_contig", "contig30", "progress_missing", "skip_after_missing"]


def _case(golden, name):
    d = os.path.join(golden, "cases", name)
    with open(os.path.join(d, "case.json")) as f:
        meta = json.load(f)
    with open(os.path.join(d, "expected.stdout"), "rb") as f:
        out = f.read()
    with open(os.path.join(d, "expected.stderr"), "rb") as f:
        err = f.read()
    return d, meta, out, err


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden_case(golden, name, tmp_path):
    d, meta, out, err = _case(golden, name)
    argv = list(meta["argv"])
    if "-p" in argv:
        argv[argv.index("-p") + 1] = str(tmp_path / "progress")
    p = _oracle.run_oracle_cli(argv, d)
    want_rc = meta["returncode"]
    if want_rc < 0:                      # the reference died of a signal (SIGSEGV): shell status 128+n
        assert p.returncode == 128 - want_rc
    else:
        assert p.returncode == want_rc
    assert p.stdout == out
    assert p.stderr == err
    if meta["progress_col1"] is not None:
        with open(tmp_path / "progress") as f:
            assert [ln.split("\t")[0].rstrip("\n") for ln in f] == meta["progress_col1"]


def test_oracle_bundled_example_md5(golden):
    """cfg 1: reference test/example.sh step 1; md5 of the 254 MB TSV recorded from the real binary."""
    b = os.path.join(golden, "bundled")
    with open(os.path.join(b, "step1_facts.json")) as f:
        facts = json.load(f)
    p = _oracle.run_oracle_cli(facts["argv"], b)
    assert p.returncode == 0
    assert len(p.stdout) == facts["stdout_bytes"]
    assert hashlib.md5(p.stdout).hexdigest() == facts["stdout_md5"] == "75989a9bc31ef0b6f53a5112a60920bd"


@pytest.mark.skipif(not _oracle.have_reference(), reason="oracle/_ref not built (GPU box)")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_equals_live_reference_on_fuzz(seed, tmp_path):
    """Random FASTA/FASTQ soup (junk bytes, CRLF, blank lines, multi-line, '>'/'@'/'+' traps)."""
    rng = random.Random(seed)
    strain = _synth.rand_dna(rng, 3000)
    lines = [b">s1 c\n"]
    for i in range(0, len(strain), 70):
        lines.append(strain[i:i + 70] + b"\n")
    (tmp_path / "strain.fa").write_bytes(b"".join(lines))
    files = []
    for fi in range(3):
        recs = _synth.fuzz_stream(rng, strain, 200, p_junk=0.01, min_len=0, max_len=160).split(b"\n")[:-1]
        body = bytearray()
        for i, r in enumerate(recs):
            style = rng.randrange(5)
            r = r.replace(b"\r", b"A")
            if style == 0:
                body += b">r%d\n" % i + r + b"\n"
            elif style == 1:
                w = rng.choice([20, 50, 61])
                body += b">r%d desc\r\n" % i + b"\r\n".join(r[j:j + w] for j in range(0, len(r), w)) + b"\r\n"
            elif style == 2:
                q = bytes(rng.choice(b"@+>IJK#") for _ in r)
                body += b"@q%d\n" % i + r + b"\n+\n" + q + b"\n"
            elif style == 3:
                body += b">r%d\n\n" % i + r[:len(r) // 2] + b"\n\n" + r[len(r) // 2:] + b"\n"
            else:
                body += b"junk before header\n>r%d\t x\n" % i + r + b"\n"
        name = "m%d.fx" % fi
        (tmp_path / name).write_bytes(bytes(body))
        files.append(name)
    (tmp_path / "A.txt").write_text(files[0] + "\n")
    (tmp_path / "B.txt").write_text("\n".join(files[1:]) + "\n")
    argv = ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt"]
    ref = _oracle.run_reference_cli(argv, str(tmp_path))
    ora = _oracle.run_oracle_cli(argv, str(tmp_path))
    assert ref.returncode == ora.returncode == 0
    assert ref.stdout == ora.stdout
    assert ref.stderr == ora.stderr


def test_oracle_on_the_cfg3_shaped_job(golden, tmp_path):
    """BASELINE configs[2] in shape -- a 1000-line -A list, a multi-file -B list (plain and .gz FASTQ), a -C list that holds
    the -r path, -p -- against what the UNMODIFIED reference printed for the same inputs (tests/golden/cfg3_shape_facts.json,
    made by tests/golden/make_cfg3_shape.py; the inputs are rebuilt here from the same seed)."""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cfg3_shape", os.path.join(golden, "make_cfg3_shape.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    facts = json.load(open(os.path.join(golden, "cfg3_shape_facts.json")))
    argv = mk.write_inputs(str(tmp_path))
    assert argv == facts["argv"]
    p = _oracle.run_oracle_cli(argv, str(tmp_path))
    assert p.returncode == facts["returncode"] and p.stderr.decode() == facts["stderr"]
    got = mk.facts_of(p.stdout, p.stderr, str(tmp_path))
    assert {k: got[k] for k in ("md5_stdout", "lines", "column_sums", "md5_progress_without_times")} == \
           {k: facts[k] for k in ("md5_stdout", "lines", "column_sums", "md5_progress_without_times")}
    if _oracle.have_reference():                     # where the reference binary exists: live, too
        q = _oracle.run_reference_cli(argv, str(tmp_path))
        assert hashlib.md5(q.stdout).hexdigest() == facts["md5_stdout"]
