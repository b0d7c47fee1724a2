"""The file rendezvous that precedes the RCCL collective (strainer2_amd/csrc/sk_rendezvous.h), several processes on the
CPU: freshness (files left by a crashed launch are ignored), agreement on a failed set-up before anyone would block,
and bounded waits.  New relative to the reference, which is single-process."""
import ctypes as C
import multiprocessing as mp
import os
import struct
import time

import pytest

import strainer2_amd.native as native

MAGIC = 0x534B5244565A3032


def _rank(rank, world, base, status, timeout, delay, q, env=None):
    os.environ.update(env or {})
    if delay:
        time.sleep(delay)
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        for i in range(128):
            buf[i] = (i * 7 + 1) & 0xFF
    rc = native.lib.sk_rendezvous_exchange(rank, world, base.encode(), status, buf, timeout)
    q.put((rank, rc, bytes(buf)))


def _run(world, base, status=None, timeout=10.0, absent=(), delays=None, timeouts=None, envs=None):
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = []
    for r in range(world):
        if r in absent:
            continue
        p = ctx.Process(target=_rank, args=(r, world, base, (status or {}).get(r, 0), (timeouts or {}).get(r, timeout),
                                            (delays or {}).get(r, 0), q, (envs or {}).get(r)))
        p.start()
        ps.append(p)
    out = {}
    for _ in ps:
        r, rc, payload = q.get(timeout=timeout + 20)
        out[r] = (rc, payload)
    for p in ps:
        p.join(timeout=10)
    return out


WANT = bytes((i * 7 + 1) & 0xFF for i in range(128))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_everyone_gets_rank0s_payload(tmp_path, world):
    out = _run(world, str(tmp_path / "id"))
    assert all(rc == 0 for rc, _ in out.values()), out
    assert all(payload == WANT for _, payload in out.values())


def test_leftovers_of_a_crashed_launch_are_ignored(tmp_path):
    base = str(tmp_path / "id")
    world = 4
    # a board and hello files with the right magic and shape, but the tokens of another launch
    board = struct.pack("<QIIQ", MAGIC, world, 0, 0) + struct.pack("<64Q", *([0x1111] * 64)) + bytes(128)
    open(base, "wb").write(board)
    for r in range(1, world):
        open(f"{base}.hello.{r}", "wb").write(struct.pack("<QIIQIIQ", MAGIC, r, world, 0x2222, 0, 0, 0))
    # rank 0 starts late: the others meet the stale board first and must not take it
    out = _run(world, base, delays={0: 0.8})
    assert all(rc == 0 for rc, _ in out.values()), out
    assert all(payload == WANT for _, payload in out.values())


@pytest.mark.parametrize("bad", [0, 2])
def test_a_failed_set_up_makes_everyone_leave(tmp_path, bad):
    out = _run(4, str(tmp_path / "id"), status={bad: 1})
    assert all(rc == 1 for rc, _ in out.values()), out          # SKR_ABORT on every rank, nobody enters the collective


def test_a_rank_that_never_shows_up_is_a_bounded_wait(tmp_path):
    t0 = time.time()
    out = _run(4, str(tmp_path / "id"), timeout=1.5, absent=(3,))
    assert time.time() - t0 < 15
    assert out[0][0] == 2                                        # rank 0 timed out ...
    assert all(rc in (1, 2) for rc, _ in out.values()), out      # ... and told the ones that did arrive to leave (or they timed out too)


def test_a_rank_that_gave_up_before_the_board_makes_everyone_leave(tmp_path):
    """rank 1 runs into its own (short) timeout and takes its hello away while rank 0 -- which has already seen
    that hello -- still waits for rank 2, who is late.  Rank 0 reads every hello once more before it publishes: the
    verdict is "leave", not a collective that is one rank short."""
    out = _run(3, str(tmp_path / "id"), timeout=8.0, timeouts={1: 0.6}, delays={2: 1.5})
    assert out[1][0] == 2                                        # rank 1 timed out on its own
    assert out[0][0] == 1 and out[2][0] == 1, out                # the others are told to leave (SKR_ABORT)


def test_two_launches_on_one_path_do_not_mix(tmp_path):
    """Every file carries the launch's nonce (MASTER_ADDR:MASTER_PORT, TORCHELASTIC_RUN_ID or SK_LAUNCH_ID): rank 1 of ANOTHER
    launch on the same default path is not taken for this launch's rank 1 -- rank 0 keeps waiting for its own."""
    base = str(tmp_path / "id")
    out = _run(2, base, timeout=1.5, envs={0: {"SK_LAUNCH_ID": "A"}, 1: {"SK_LAUNCH_ID": "B"}})
    assert out[0][0] == 2 and out[1][0] == 2, out                # neither accepts the other: both time out
    out = _run(2, base, timeout=5.0, envs={0: {"SK_LAUNCH_ID": "A"}, 1: {"SK_LAUNCH_ID": "A"}})
    assert out[0][0] == 0 and out[1][0] == 0 and out[1][1] == WANT


def test_without_rank0_the_others_time_out(tmp_path):
    out = _run(3, str(tmp_path / "id"), timeout=1.0, absent=(0,))
    assert all(rc == 2 for rc, _ in out.values()), out
    assert not [f for f in os.listdir(tmp_path) if "hello" in f]   # their hello files are gone


def test_program_ranks_leave_together_when_a_set_up_fails(tmp_path):
    """bin/kmer_scrub_count started as two ranks: here at least one rank's set-up fails (no usable device in the build
    container, or -- on a one-GPU box -- no second device for rank 1; rank 1's strain is unreadable as well).  Through
    sk_comm_init_ex every rank learns it BEFORE anyone would enter ncclCommInitRank, prints its own reason and exits
    non-zero within the rendezvous -- no hang, no leftover files that would confuse the next launch."""
    import subprocess
    import strainer2_amd as sk
    (tmp_path / "s.fa").write_bytes(b">s\n" + b"ACGTTGCA" * 40 + b"\n")
    (tmp_path / "A.txt").write_text("")
    (tmp_path / "B.txt").write_text("")
    idf = str(tmp_path / "rccl_id")
    ps = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), SK_RCCL_ID_FILE=idf, SK_RENDEZVOUS_TIMEOUT="20")
        strain = str(tmp_path / ("s.fa" if r == 0 else "missing.fa"))
        ps.append(subprocess.Popen([sk.cli_path(), "-r", strain, "-A", str(tmp_path / "A.txt"), "-B", str(tmp_path / "B.txt")],
                                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=60) for p in ps]
    assert time.time() - t0 < 30
    assert all(p.returncode != 0 for p in ps), [(p.returncode, o[1][-300:]) for p, o in zip(ps, outs)]
    assert outs[0][0] == b"" and outs[1][0] == b""                       # nobody printed a table
    assert b"could not read file" in outs[1][1]                           # rank 1 says why
    assert not [f for f in os.listdir(tmp_path) if "hello" in f]
