"""`python bench.py --gpus N` with no launcher around it must start its own N ranks (the driver's command is
the plain one, and round 3's bench.py answered it with "launch with torch.distributed.run").  The parent picks a free port,
runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD, passes exactly
one JSON line on and returns the child's status -- and never imports torch or the library itself (a process that has
initialised the GPU must not start others by exec; this one starts them without ever touching it).  The ranks here are a
stub (tests/native/bench_rank_stub.py, gloo): the real ranks need a GPU; tests/test_scale_checks_gpu.py runs those."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(REPO, "tests", "native", "bench_rank_stub.py")


def _run(argv, **env):
    e = dict(os.environ, SK_BENCH_RANK_SCRIPT=STUB, **env)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=e, capture_output=True, timeout=300)


def test_plain_command_starts_the_ranks_and_relays_one_line():
    p = _run(["--gpus", "2", "--steps", "7", "--warmup", "2", "--backend", "gloo"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1                                                  # the noise of both ranks went to stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] == 3 and line["steps"] == 7 and line["warmup"] == 2 and int(line["port"]) > 0
    err = p.stderr.decode()
    assert "noise on stdout from rank 0" in err and "noise on stdout from rank 1" in err
    assert "--nproc-per-node=2 --master-addr 127.0.0.1 --master-port %s" % line["port"] in err


def test_a_failing_rank_fails_the_plain_command():
    p = _run(["--gpus", "2", "--steps", "99", "--backend", "gloo"])
    assert p.returncode != 0 and p.stdout == b""


def test_the_parent_touches_neither_torch_nor_the_library():
    code = ("import runpy, sys\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--backend', 'gloo']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.') or m.startswith('strainer2_amd')]\n"
            "assert not bad, bad\nprint('parent clean', file=sys.stderr)\n" % os.path.join(REPO, "bench.py"))
    e = dict(os.environ, SK_BENCH_RANK_SCRIPT=STUB)
    e.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, timeout=300)
    assert p.returncode == 0 and b"parent clean" in p.stderr, p.stderr.decode()[-2000:]


def test_under_a_launcher_the_world_must_match():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0", SK_BENCH_RANK_SCRIPT=STUB)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-cpu"], env=e, capture_output=True, timeout=120)
    assert p.returncode != 0 and b"must be the same" in p.stderr
