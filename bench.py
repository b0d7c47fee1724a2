#!/usr/bin/env python3
"""bench.py -- the k-mer scrub/count hot path on N MI355X, one process per GPU.

Metric (BASELINE.json): metagenome bases/sec scanned at k=31, bit-exact k-mer counts.
Workload at every N (weak scaling): BASELINE configs[1] per GPU -- one 5 Mbp synthetic strain
(-r) resident as the key table, 10 M x 150 bp synthetic reads (-B; 1.5 Gbase, SURVEY 8(d)
recipe, seed 0x5EED31 + 1 + rank) resident in HBM as a record stream.  A "step" = one pass of the
scan over that batch (sk_scan_device -> sk_scan_grid [+ sk_scan_wide early-exit]).  After the
K timed steps the per-k-mer count vector of the scanned (-B) column is summed across ranks with one
RCCL all-reduce (inside the timed region when N > 1).  value = bases all ranks scanned / max-over-ranks time.

Bit-exactness is checked in the run that is timed: the counts of the K timed passes, divided by K, must be
the vector the UNMODIFIED reference program produced for the same stream (tests/golden/cfg2_facts.json:
sum, non-zero rows and md5 of the u32 column in the reference's row order, made by
tests/golden/make_cfg2_facts.py); for N > 1 the reduced total must be the sum of the ranks' facts and every
rank's own stream is checked once more by one untimed pass.

roofline: bound = HBM.  achieved = COMPULSORY bytes per launch / average kernel time, where compulsory =
the record stream read once (bases + separators) + 8 B per hit (the u32 counter's read-modify-write);
frac = achieved / 8 TB/s.  SURVEY 8(d)'s model figure (7.4 B/base: one 8-byte probe per window) is carried
beside it as `survey_model_*`: this kernel answers 16 windows with one filter block and never moves those
bytes, so that figure is not a fraction of anything.  traffic = measured HBM bytes per launch from the PMC
passes of tools/profile.sh, taken from profiles/traffic.json ONLY if that file was measured on this very
sk_device.hip + sk_dev_*.hip.h (sha256 stamp), else null.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--no-cpu]
  N > 1, either way:
    python bench.py --gpus N ...                       (no WORLD_SIZE in the environment: this process starts the N ranks itself --
                                                        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
                                                        127.0.0.1 --master-port <a free one> bench.py --gpus N ...` as a CHILD process,
                                                        never touching the GPU itself -- passes rank 0's JSON line on and exits with
                                                        the child's status)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (already under a launcher: a rank)
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import hashlib

SURVEY_MODEL_BYTES_PER_BASE = 7.4     # 1 B base + 8 B key probe x 0.8 windows/base (a model of a per-window-probe
                                      # algorithm; reported as a side field only)
HIT_BYTES = 8.0                       # u32 counter read-modify-write per hit
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_REQ_CEILING = 270e9            # requests/s the L2 takes for random 8-byte loads from an L2-resident table, measured on this chip
                                  # (profiles/r01_gather_microbench.txt): one request per channel clock over 8 XCDs x 16 channels
VALU_CYCLES_PER_WAVE_INSTR = 4    # a wave64 vector instruction occupies its 16-lane SIMD for four cycles (CDNA)


# ----------------------------------------------------------------------------- CPU baseline
_ORACLE_TABLE = None


def _cpu_worker(args):
    shard, col = args
    t0 = time.perf_counter()
    _ORACLE_TABLE.scan_stream(shard, col)
    return time.perf_counter() - t0


def host_cpus():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota where there is one
    (the GPU boxes show 256 CPUs and grant 16 CPUs' worth of time), at most 64"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(sstream, reads, read_len, target_seconds=12.0):
    """The oracle ("port": same string-keyed work per window as the reference) timed on this host's
    cores on a bounded sample of the same reads.  P forked workers share one built table."""
    global _ORACLE_TABLE
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import _oracle
    cores = host_cpus()
    t = _oracle.OracleTable()
    tb = time.perf_counter()
    assert t.build_stream(sstream) == 0
    build_s = time.perf_counter() - tb
    _ORACLE_TABLE = t
    rec = read_len + 1
    # calibrate on 20 k reads, then size the sample for ~target_seconds per core
    cal = reads[: 20_000 * rec].tobytes()
    t0 = time.perf_counter()
    t.scan_stream(cal, 3)
    rate1 = 20_000 * read_len / (time.perf_counter() - t0)
    per_core_reads = int(min(max(rate1 * target_seconds / read_len, 20_000), (reads.size // rec) // cores))
    shards = [(reads[i * per_core_reads * rec:(i + 1) * per_core_reads * rec].tobytes(), 3) for i in range(cores)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, shards)
    wall = time.perf_counter() - t0
    bases = cores * per_core_reads * read_len
    _ORACLE_TABLE = None
    t.close()
    return {"value": bases / wall, "unit": "bases/s", "cores": cores, "kind": "port",
            "sample": f"{cores} forked workers x {per_core_reads} reads of the same synthetic stream "
                      f"({bases / 1e6:.0f} Mbase, {wall:.1f} s wall); 1-core rate {rate1 / 1e6:.2f} Mbase/s; "
                      f"oracle string-table build {build_s:.1f} s (not counted)"}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_reference(contigs, reads, read_len, reads_per_core=1_000_000):
    """The real reference program (oracle/_ref/kmer_scrub_count, built by oracle/Makefile from the unmodified
    sources where /root/reference exists; the binary travels with the repo snapshot) on this host's cores:
    P processes, each scanning its own FASTA slice of the same synthetic reads against the same strain --
    up to 1 M reads per core, i.e. with 10 or more cores the WHOLE cfg-2 stream (1.5 Gbase; SURVEY 8(d) asks for a
    1 Gbase subsample).  The strain build + table print that every process also does is timed separately, TWICE
    (P processes with an empty -B list, before and after), and the mean subtracted; both readings are reported.
    None if the binary is not there."""
    import shutil
    import subprocess
    import tempfile
    from strainer2_amd import synth
    exe = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")
    if not os.access(exe, os.X_OK):
        return None
    cores = host_cpus()
    rec = read_len + 1
    per = int(min(reads_per_core, (reads.size // rec) // cores))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None     # (no disk in the timing)
    work = tempfile.mkdtemp(prefix="sk_cpu_ref_", dir=base)
    try:
        with open(os.path.join(work, "strain.fa"), "wb") as f:
            f.write(synth.strain_fasta(contigs))
        open(os.path.join(work, "empty.txt"), "w").close()
        head = np.frombuffer(b">r\n", dtype=np.uint8)
        for i in range(cores):
            rows = reads[i * per * rec:(i + 1) * per * rec].reshape(per, rec)
            fa = np.empty((per, 3 + rec), dtype=np.uint8)
            fa[:, :3] = head
            fa[:, 3:] = rows
            fa.tofile(os.path.join(work, f"reads{i}.fa"))
            with open(os.path.join(work, f"B{i}.txt"), "w") as f:
                f.write(os.path.join(work, f"reads{i}.fa") + "\n")

        def run_all(lists):
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as null:
                ps = [subprocess.Popen([exe, "-r", os.path.join(work, "strain.fa"), "-A", os.path.join(work, "empty.txt"), "-B", b],
                                       stdout=null, stderr=null) for b in lists]
                rcs = [p.wait() for p in ps]
            assert all(rc == 0 for rc in rcs), "reference program failed"
            return time.perf_counter() - t0

        fixed1 = run_all([os.path.join(work, "empty.txt")] * cores)
        full = run_all([os.path.join(work, f"B{i}.txt") for i in range(cores)])
        fixed2 = run_all([os.path.join(work, "empty.txt")] * cores)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    bases = cores * per * read_len
    fixed = 0.5 * (fixed1 + fixed2)
    if full - fixed > 0.4 * full:
        scan_s, how = full - fixed, (f"minus {fixed:.2f} s = the mean of two runs ({fixed1:.2f} s before, {fixed2:.2f} s after) of {cores} "
                                     f"concurrent strain builds + table prints with an empty -B list")
    else:                                                # sample too small for the subtraction to mean anything
        scan_s, how = full, f"strain build + table print ({fixed1:.2f} / {fixed2:.2f} s with an empty -B list) NOT subtracted"
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()[0]
    except OSError:
        quota = "?"
    return {"value": bases / scan_s, "unit": "bases/s", "cores": cores, "kind": "reference",
            "per_core_bases_per_s": bases / scan_s / cores, "cpu_model": cpu_model(),
            "cpus_visible": os.cpu_count(), "cpus_granted": cores,
            "cores_note": f"{cores} = the CPUs this job may use (affinity mask cut to the cgroup quota, cpu.max {quota}): a share of the host, "
                          f"not a whole socket of {os.cpu_count()} CPUs; the reference is single-threaded, so P independent processes (SURVEY 8(d) protocol b)",
            "full_run_seconds": full, "fixed_cost_seconds": [fixed1, fixed2], "sample_bases": bases,
            "sample": f"the unmodified reference kmer_scrub_count, {cores} processes x {per} reads of the same synthetic stream as FASTA "
                      f"({bases / 1e6:.0f} Mbase): {full:.1f} s wall, {how}; {bases / scan_s / cores / 1e6:.2f} Mbase/s per core"}


# ----------------------------------------------------------------------------- facts and measured traffic
def load_cfg2_facts(args):
    """the committed facts of the reference's run on the cfg-2 stream, if this run IS cfg 2"""
    p = os.path.join(REPO, "tests", "golden", "cfg2_facts.json")
    if not os.path.exists(p) or args.strain_bp != 5_000_000 or args.hit_frac != 0.02:
        return None
    f = json.load(open(p))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if f.get("reads") != args.reads or f.get("read_len") != args.read_len or len(f.get("ranks", [])) < world:
        return None
    return f


def device_source_sha():
    """sha256 over the device translation unit: sk_device.hip and the sk_dev_*.hip.h parts it includes (names and bytes, in name order)"""
    d = os.path.join(REPO, "strainer2_amd", "csrc")
    h = hashlib.sha256()
    for name in ["sk_device.hip"] + sorted(n for n in os.listdir(d) if n.startswith("sk_dev_") and n.endswith(".hip.h")):
        h.update(name.encode() + b"\0")
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_traffic(args, kernel_name):
    """HBM bytes per launch from the PMC passes (tools/profile.sh -> tools/save_profile.py), only if they were
    taken on this very kernel source and workload; (None, why) otherwise"""
    tp = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(tp):
        return None, "no profiles/traffic.json"
    try:
        tj = json.load(open(tp))
    except Exception as e:                                   # noqa: BLE001
        return None, f"unreadable profiles/traffic.json ({e})"
    if tj.get("reads") != args.reads or tj.get("kernel") != kernel_name:
        return None, "profiles/traffic.json is for another workload/kernel"
    if tj.get("sk_device_hip_sha256") != device_source_sha():
        return None, "profiles/traffic.json was measured on another version of sk_device.hip / sk_dev_*.hip.h (stale): re-run tools/profile.sh"
    return tj.get("hbm_bytes_per_launch"), tj.get("correction")


def measured_vector_issue():
    """vector-instruction issue of the scan kernel as a fraction of the SIMDs' issue slots, from the same stamped PMC passes:
    SQ_INSTS_VALU x 4 cycles (a wave64 instruction on a 16-lane SIMD) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); None if stale"""
    tp = os.path.join(REPO, "profiles", "traffic.json")
    try:
        tj = json.load(open(tp))
        if tj.get("sk_device_hip_sha256") != device_source_sha():
            return None
        n, cyc = tj.get("vector_instructions_per_launch"), tj.get("gpu_cycles_per_launch_per_xcd")
        if not n or not cyc:
            return None
        return {"instructions_per_launch": n, "cycles_per_launch": cyc, "frac": n * 4.0 / 1024.0 / cyc,
                "what": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), profiled run of the same source (profiles/traffic.json)"}
    except Exception:                                        # noqa: BLE001
        return None


# ----------------------------------------------------------------------------- strain_detect side measurement
def strain_detect_leg(device, oracle_sample_reads=20_000, repeat=10):
    """The second deliverable's kernel (sk_scan_grid<TALLY, UNION>: per-read tallies against ONE table over 32 resident strains,
    BASELINE configs[4] at one GPU's share: 256 strains / 8) on resident batches of 32 MiB and 512 MiB of 150-base reads, 2 % of them
    cut from the strains, 1 % of every strain's rows informative.  Reported beside `value`, never as it: kernel ms from HIP events
    on the union's stream (sk_union_scan_timing), wall per launch + collect round, roofline fraction on compulsory bytes (the batch
    read once + 16 B per (read, strain) pair that was hit + 8 B per log entry).  Checked in the run: the union's results ==
    member by member (32 x sk_tally_launch on the strains' own tables, record for record and log entry for log entry), and for two
    strains the sums over a sample of reads == the oracle's counts for that sample (all hits; hits on informative rows)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import strainer2_amd as sk
    from strainer2_amd import cfg5
    from strainer2_amd.native import lib
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import _oracle
    ns = cfg5.NSTRAINS
    t0 = time.perf_counter()
    strains = [cfg5.strain(s) for s in range(ns)]
    with ThreadPoolExecutor(min(8, host_cpus())) as ex:                       # (the key-set build releases the GIL)
        sets = list(ex.map(lambda g: sk.Keyset.from_stream(g.tobytes() + b"\n", default_val=1, incr=0), strains))
    t_sets = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctxs, types = [], []
    for s in range(ns):
        c = sk.KmerContext(device)
        c.load_keyset(sets[s], 6)
        typ = np.ones(sets[s].nrows, dtype=np.uint32)
        typ[np.random.default_rng(cfg5.SEED_I + s).choice(sets[s].nrows, sets[s].nrows // 100, replace=False)] = 2
        c.set_counts(0, typ)
        ctxs.append(c)
        types.append(typ)
    for c in ctxs:
        c.sync()
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    u = sk.KmerUnion(ctxs, 0, 2)
    t_union = time.perf_counter() - t0
    genome = np.concatenate(strains)
    out = {"strains": ns, "strain_bp": cfg5.STRAIN_BP, "union_rows": int(u.rows), "strain_read_fraction": 0.02, "informative_row_fraction": 0.01,
           "key_sets_s": round(t_sets, 2), "table_loads_s": round(t_load, 2), "union_build_s": round(t_union, 3), "batches": []}
    rec = cfg5.READ_LEN + 1
    for mib in (32, 512):
        nrec = (mib << 20) // rec
        stream = np.empty((nrec, rec), dtype=np.uint8)
        for a in range(0, nrec, cfg5.BLOCK):
            m = min(cfg5.BLOCK, nrec - a)
            stream[a:a + m, :] = cfg5.reads_block(1000 + a // cfg5.BLOCK, m, genome)[:, 3:]       # (the FASTA block without its header lines)
        starts = (np.arange(nrec, dtype=np.uint64) * rec).astype(np.uint32)
        nbytes = nrec * rec
        assert lib.sk_batch_fill(u._batch, stream.ctypes.data, nbytes, starts.ctypes.data, nrec) == 0
        cap = max(nrec // 2, 1 << 16)
        recs = np.zeros((min(nrec * ns, 1 << 26) + 1, 3), dtype=np.uint32)
        hits = np.zeros((cap, 2), dtype=np.uint32)
        nr, nh = C.c_uint64(0), C.c_uint64(0)

        def one_round():
            assert lib.sk_union_tally_launch(u._h, u._batch, cap) == 0, lib.sk_union_last_error(u._h)
            assert lib.sk_union_tally_collect(u._h, recs.ctypes.data, len(recs) - 1, C.byref(nr), hits.ctypes.data, C.byref(nh)) == 0
            assert nh.value <= cap and nr.value < len(recs)
        one_round()                                                            # warm-up (scratch buffers grow here)
        lib.sk_union_scan_timing(u._h, None, None, 1)
        t0 = time.perf_counter()
        for _ in range(repeat):
            one_round()
        wall = (time.perf_counter() - t0) / repeat
        ms, nl = C.c_double(0), C.c_uint64(0)
        assert lib.sk_union_scan_timing(u._h, C.byref(ms), C.byref(nl), 1) == 0
        kern_ms = ms.value / max(nl.value, 1)
        pairs, nlog = int(nr.value), int(nh.value)
        got_recs = recs[:pairs].copy()
        got_hits = hits[:nlog].copy()
        compulsory = nbytes + 16.0 * pairs + 8.0 * nlog
        entry = {"batch_mib": mib, "reads": nrec, "bases": nrec * cfg5.READ_LEN, "kernel": "sk_scan_grid<TALLY,UNION>", "kernel_ms": kern_ms,
                 "launches_timed": int(nl.value), "wall_ms_launch_and_collect": wall * 1e3,
                 "bases_per_s_kernel": nrec * cfg5.READ_LEN / (kern_ms * 1e-3), "strain_x_bases_per_s_kernel": ns * nrec * cfg5.READ_LEN / (kern_ms * 1e-3),
                 "read_strain_pairs_hit": pairs, "log_entries": nlog,
                 "roofline": {"bound": "hbm", "compulsory_bytes": compulsory, "achieved": compulsory / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": compulsory / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if mib == 32:
            # ---- union == member by member, record for record and log entry for log entry
            dense = np.zeros((nrec * ns, 2), dtype=np.uint32)
            dense[got_recs[:, 0]] = got_recs[:, 1:]
            dense = dense.reshape(nrec, ns, 2)
            one = np.zeros((nrec + 1, 3), dtype=np.uint32)
            mh = np.zeros((cap, 2), dtype=np.uint32)
            same = True
            for s, c in enumerate(ctxs):
                assert lib.sk_tally_launch(c._h, u._batch, 0, 2, cap) == 0
                assert lib.sk_tally_collect_sparse(c._h, one.ctypes.data, nrec, C.byref(nr), mh.ctypes.data, C.byref(nh)) == 0
                md = np.zeros((nrec, 2), dtype=np.uint32)
                md[one[:nr.value, 0]] = one[:nr.value, 1:]
                sel = got_hits[(got_hits[:, 1] >> 27) == s]
                mine = np.stack([sel[:, 0], sel[:, 1] & ((1 << 27) - 1)], axis=1)
                theirs = mh[:nh.value]
                same = same and np.array_equal(md, dense[:, s, :]) and \
                    np.array_equal(mine[np.lexsort((mine[:, 1], mine[:, 0]))], theirs[np.lexsort((theirs[:, 1], theirs[:, 0]))])
            assert same, "union and member-by-member tallies differ"
            entry["union_equals_member_by_member"] = True
            # ---- two strains against the oracle on a sample of reads
            sample = stream[:oracle_sample_reads].tobytes()
            oc = {}
            for s in cfg5.PINNED_STRAINS:
                t = _oracle.OracleTable(ncols=6)
                assert t.build_stream(strains[s].tobytes() + b"\n", default=1, incr=0, short_policy=1) == 0
                t.scan_stream(sample, 1)
                okeys, ocounts = t.rows()
                assert okeys == sets[s].keys(), "row order differs from the oracle"
                col = ocounts[:, 1].astype(np.int64)
                want = (int(col.sum()), int(col[types[s] == 2].sum()))
                got = (int(dense[:oracle_sample_reads, s, 0].sum()), int(dense[:oracle_sample_reads, s, 1].sum()))
                assert want == got, f"strain {s}: tallies of the sample differ from the oracle ({got} vs {want})"
                oc[str(s)] = {"all_hits": got[0], "informative_hits": got[1]}
                t.close()
            entry["oracle_sample"] = {"reads": oracle_sample_reads, "strains": oc, "equal": True}
        out["batches"].append(entry)
        del stream
    u.close()
    for c in ctxs:
        c.close()
    return out


# ----------------------------------------------------------------------------- N > 1 from the plain command
def launch_ranks(gpus, argv):
    """`python bench.py --gpus N` with no launcher around it (the driver's own command): start the N ranks as a
    child `python -m torch.distributed.run ...` on a free port, relay rank 0's JSON line (stdout lines that are not the line go to
    stderr: RCCL/torch banners must not break the one-line contract) and return the child's exit status.  This parent imports
    neither torch nor the library: a process that has initialised the GPU must not start others by exec, and it has no need to.
    SK_BENCH_RANK_SCRIPT (tests only): the script the ranks run instead of this file."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.environ.get("SK_BENCH_RANK_SCRIPT") or os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: what RCCL needs between processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: starting %d ranks: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    lines = 0
    for raw in child.stdout:
        text = raw.decode("utf-8", "replace")
        is_line = False
        if text.lstrip().startswith("{"):
            try:
                is_line = "metric" in json.loads(text)
            except ValueError:
                is_line = False
        if is_line:
            lines += 1
            sys.stdout.write(text if text.endswith("\n") else text + "\n")
            sys.stdout.flush()
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if rc == 0 and lines != 1:
        print("bench.py: the ranks ended with status 0 but printed %d result lines" % lines, file=sys.stderr)
        rc = 1
    return rc if rc >= 0 else 128 - rc


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed passes (1000 x 0.8 ms: long enough for an outside GPU-busy sampler to see)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed launches of the same scan before the W warmup steps, for this long: the card's clocks ramp over the first ~100 "
                         "launches (20 timed steps straight after start: 0.648 ms a launch; after 0.3 s of launches: 0.610) -- reported as preheat_* in the line; 0 = none")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (cfg 2: 10 M)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--strain-bp", type=int, default=5_000_000, help="size of the synthetic strain (cfg 2: 5 Mbp); other sizes are for sweeps, not for `value`")
    ap.add_argument("--hit-frac", type=float, default=0.02, help="fraction of reads drawn from the strain (cfg 2: 0.02)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-sd", action="store_true", help="skip the strain_detect (TALLY/UNION kernel) side measurement")
    ap.add_argument("--sd-only", action="store_true", help="only the strain_detect side leg (for tools/profile.sh --sd): prints its object, no `value`")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the PCIe-inclusive host-buffer passes (keeps profiles clean)")
    ap.add_argument("--file-reads", type=int, default=1_000_000, help="reads per FASTQ file of the file-fed, rank-sharded side measurement (0 = skip it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--grid-kib", type=int, default=None, help="size of the grid kernel's level-1 filter in KiB (default: automatic)")
    ap.add_argument("--text-stage", type=int, default=None, help="0 = stage 2 without the strain's text (A/B)")
    ap.add_argument("--ablate", type=int, default=0, help="timing experiments: 1 = no filter/table memory, 2 = no table probes (counts are wrong)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # the plain command: be the launcher (no GPU, no torch in this process)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world and --gpus must be the same")

    if args.sd_only:                                            # (profiles of sk_scan_grid<TALLY,UNION>: nothing else on the card)
        if world != 1:
            raise SystemExit("--sd-only is a single-GPU measurement")
        print(json.dumps({"metric": "strain_detect side leg only (not the headline metric)", "value": None, "n_gpus": 1,
                          "strain_detect": strain_detect_leg(0)}), flush=True)
        return

    from strainer2_amd import synth
    contigs = synth.make_strain(total_bp=args.strain_bp)
    sstream = synth.strain_stream(contigs)
    reads, nbases = synth.make_reads(contigs, args.reads, args.read_len, hit_frac=args.hit_frac, seed=synth.SEED + 1 + rank)

    # CPU baseline first: it forks, and must do so before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline_reference(contigs, reads, args.read_len)
        port = cpu_baseline(sstream, reads, args.read_len, target_seconds=1.0 if cpu else 12.0)
        if cpu is None:
            cpu = port                                   # no reference binary here: the oracle ("port") stands in
        else:
            cpu["port_bases_per_s"] = port["value"]      # the oracle restatement on the same cores, for continuity
            cpu["port_sample"] = port["sample"]

    import torch
    import torch.distributed as dist
    import strainer2_amd as sk

    device = local_rank % max(torch.cuda.device_count(), 1)      # (rehearsal: more ranks than GPUs share a card)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ks = sk.Keyset.from_stream(sstream)
    ctx = sk.KmerContext(device)
    if args.grid_kib is not None:
        ctx.set_option("grid_kib", args.grid_kib)
    if args.text_stage is not None:
        ctx.set_option("text_stage", args.text_stage)
    if args.ablate:
        ctx.set_option("ablate", args.ablate)
    ctx.load_keyset(ks, 4)
    dev = ctx.dev_alloc(reads.size)
    ctx.dev_upload(dev, reads)
    nbytes = int(reads.size)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from strainer2_amd.dist import allreduce_counts

    preheat_launches = 0
    if args.preheat_ms > 0:                 # bring the card to its running clocks (untimed; the counters are zeroed below)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
            for _ in range(20):
                ctx.scan_device(dev, nbytes, 2)
            ctx.sync()
            preheat_launches += 20
    for _ in range(args.warmup):
        ctx.scan_device(dev, nbytes, 2)
    if world > 1:
        allreduce_counts(ctx, 2)            # warm the communicator up
    ctx.zero_counts(2)
    barrier()
    ctx.scan_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.scan_device(dev, nbytes, 2)
    if world > 1:
        allreduce_counts(ctx, 2)            # the -B column's per-k-mer count vector (20 MB at cfg 2)
    else:
        ctx.counts_device_ptr()             # folds the pending run-length increments into the column: part of the timed work
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = ctx.scan_timing(reset=True)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # bit-exactness inside the same run: the K timed passes' counts / K must be the reference program's vector
    # for this very stream (tests/golden/cfg2_facts.json); for N > 1 the reduced total must be the ranks' sum
    counts = ctx.counts(2)
    if world > 1:                                   # every rank must hold the same reduced table
        chk = torch.tensor([int(counts.astype(np.uint64).sum())], dtype=torch.int64, device="cuda")
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert int(lo.item()) == int(hi.item()), "ranks disagree after the all-reduce"
    assert args.ablate or (int(counts.sum()) % args.steps == 0 and np.all(counts % args.steps == 0)), "counts not K x one pass"
    hits_per_pass = int(counts.astype(np.uint64).sum()) // args.steps
    parity = {"facts": None, "checked": False}
    facts = load_cfg2_facts(args)
    if facts is not None and not args.ablate:
        def fact_of(vec):
            return {"sum": int(vec.astype(np.uint64).sum()), "nonzero_rows": int(np.count_nonzero(vec)),
                    "md5_u32_le": hashlib.md5(vec.astype("<u4").tobytes()).hexdigest()}
        if world == 1:
            got, want = fact_of(counts // args.steps), facts["ranks"][0]
            assert all(got[k] == want[k] for k in got), f"timed run differs from the reference: {got} vs {want}"
        else:
            want_sum = sum(f["sum"] for f in facts["ranks"][:world])
            assert hits_per_pass == want_sum, f"reduced total {hits_per_pass} != the ranks' reference sum {want_sum}"
            ctx.zero_counts(3)
            ctx.scan_device(dev, nbytes, 3)                     # one untimed pass of this rank's own stream
            got, want = fact_of(ctx.counts(3)), facts["ranks"][rank]
            ok = torch.tensor([int(all(got[k] == want[k] for k in got))], dtype=torch.int64, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            assert int(ok.item()) == 1, f"a rank's pass differs from the reference (rank {rank}: {got} vs {want})"
        parity = {"facts": "tests/golden/cfg2_facts.json", "checked": True, "producer": facts.get("producer"),
                  "what": "counts of the timed passes / steps == the reference's metagenome_count column (sum, non-zero rows, md5)"
                          if world == 1 else "reduced total == sum of the ranks' reference sums; every rank's own stream re-scanned once and md5-checked"}

    # PCIe-inclusive rate (host buffer -> pinned staging -> H2D -> scan), reported beside `value`,
    # never as it: 2 passes of sk_scan_stream over the same record stream
    host_rate = pinned_rate = packed_resident = None
    if world == 1 and not args.ablate and not args.no_host_rate:
        ctx.scan_stream(reads[: 64 << 20], 3)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(2):
            ctx.scan_stream(reads, 3)
        ctx.sync()
        host_rate = 2 * nbases / (time.perf_counter() - t1)
        # the same from PINNED host memory (what the program's decode threads fill): DMA in place
        rec = args.read_len + 1
        chunk_reads = (48 << 20) // rec
        pin = ctx.pinned_alloc(reads.size)
        pin[:] = reads
        t1 = time.perf_counter()
        for _ in range(2):
            for a in range(0, args.reads, chunk_reads):
                n = min(chunk_reads, args.reads - a)
                ctx.scan_pinned(pin, n * rec, 3, offset=a * rec)
        ctx.sync()
        pinned_rate = 2 * nbases / (time.perf_counter() - t1)
        ctx.pinned_free(pin)
        # the same stream in the host-PACKED form (sk_pack_stream: 6 bytes per 16 bases; what the list scans upload): resident on the
        # device, the kernel's first phase only copies -- another input form than `value`'s, reported beside it, counts checked
        if args.hit_frac == 0.02:
            pk, odd = sk.pack_stream(reads)
            if not odd:
                dpk = ctx.dev_alloc(pk.size)
                ctx.dev_upload(dpk, pk)
                ctx.zero_counts(3)
                ctx.scan_device_packed(dpk, nbytes, 3)
                ctx.sync()
                assert np.array_equal(ctx.counts(3) * args.steps, counts), "the packed form counts something else"
                ctx.scan_timing(reset=True)
                for _ in range(50):
                    ctx.scan_device_packed(dpk, nbytes, 3)
                ctx.sync()
                pms, pn = ctx.scan_timing(reset=True)
                packed_resident = {"kernel": "sk_scan_grid<PACKED>", "avg_launch_ms": pms / max(pn, 1), "launches": int(pn),
                                   "bases_per_s": nbases / (pms / max(pn, 1) * 1e-3), "input_bytes_per_launch": int(pk.size),
                                   "what": "the same reads resident in the host-packed form (a code word and a mask per 16-byte chunk, 6 bytes instead of 16): "
                                           "the kernel's phase 1 copies instead of decoding; counts == the byte form's, checked.  NOT the input `value` is quoted on"}
                ctx.dev_free(dpk)

    # File-fed, rank-sharded side measurement (never `value`): every rank writes 2 FASTQ files of its own reads under
    # /dev/shm (or TMPDIR), ONE list names them all plus one file four times the size, and every rank scans its share of
    # that list through the product's list walk (skh_scan_list: items dealt by size, the big file cut at checked record
    # boundaries, decode threads -> pinned buffers -> H2D -> kernel).  This is where N ranks share the host's cores.
    file_fed = None
    if args.file_reads > 0 and not args.ablate:
        import shutil
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        root = os.path.join(base or tempfile.gettempdir(), "sk_bench_files_%s" % os.environ.get("MASTER_PORT", str(os.getuid())))
        rec = args.read_len + 1
        n_small = min(args.file_reads, args.reads // 8)

        def everyone_ok(ok, why):
            """a side measurement must never hang the headline run: every step that can fail locally (disk full, a file
            missing) is followed by an agreement, and the leg is dropped on EVERY rank if it failed on one"""
            if world > 1:
                t_ok = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                ok = bool(t_ok.item())
            return ok, (None if ok else {"skipped": why})

        def write_fastq(path, rows):
            n = rows.shape[0]
            fq = np.empty((n, 3 + rec + 2 + rec), dtype=np.uint8)
            fq[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
            fq[:, 3:3 + rec] = rows
            fq[:, 3 + rec:5 + rec] = np.frombuffer(b"+\n", dtype=np.uint8)
            fq[:, 5 + rec:5 + rec + args.read_len] = ord("I")
            fq[:, 5 + rec + args.read_len] = ord("\n")
            fq.tofile(path)

        ok, why = True, ""
        try:
            os.makedirs(root, exist_ok=True)
            for k in range(2):
                write_fastq(os.path.join(root, f"r{rank}_{k}.fq"), reads[k * n_small * rec:(k + 1) * n_small * rec].reshape(n_small, rec))
            if rank == 0:
                write_fastq(os.path.join(root, "big.fq"), reads[:4 * n_small * rec].reshape(4 * n_small, rec))
                with open(os.path.join(root, "list.txt"), "w") as f:
                    f.write(os.path.join(root, "big.fq") + "\n")
                    for r in range(world):
                        for k in range(2):
                            f.write(os.path.join(root, f"r{r}_{k}.fq") + "\n")
        except Exception as e:                               # noqa: BLE001
            ok, why = False, f"could not write the files: {e}"
        ok, file_fed = everyone_ok(ok, "a rank could not write its FASTQ files under " + root + (": " + why if why else ""))
        if ok:
            # dist.scan_list_sharded: the plans are compared, a cut that does not hold sends EVERY rank round again uncut, and any
            # failure raises on every rank together (the same collectives on all ranks whatever happens to one of them)
            from strainer2_amd.dist import scan_list_sharded
            cold = None
            try:                                             # first pass, not the one reported: page-locks the decode threads' buffers
                t1 = time.perf_counter()
                scan_list_sharded(ctx, os.path.join(root, "list.txt"), 1, rank, world)
                ctx.sync()
                cold = time.perf_counter() - t1
            except (OSError, ValueError) as e:
                ok, why = False, f"scan_list failed: {e}"
            barrier()
            ctx.zero_counts(1)
            t1 = time.perf_counter()
            fb = 0
            FILE_PASSES = 3                                  # (one pass of this small list is 60 ms: three make a steadier number)
            try:
                for _ in range(FILE_PASSES if ok else 0):
                    fb += scan_list_sharded(ctx, os.path.join(root, "list.txt"), 1, rank, world)
                ctx.sync()
            except (OSError, ValueError) as e:
                ok, why = False, f"scan_list failed: {e}"
            if world > 1:
                allreduce_counts(ctx, 1)
            barrier()
            dt = time.perf_counter() - t1
            fb_total, dt_max = fb, dt
            if world > 1:
                tt = torch.tensor([float(fb), dt], dtype=torch.float64, device="cuda")
                tsum, tmax2 = tt.clone(), tt.clone()
                dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                dist.all_reduce(tmax2, op=dist.ReduceOp.MAX)
                fb_total, dt_max = float(tsum[0].item()), float(tmax2[1].item())
            ok, file_fed = everyone_ok(ok, "a rank's list scan failed" + (": " + why if why else ""))
            if ok:
                file_fed = {"bases_per_s": fb_total / dt_max, "bases": int(fb_total), "seconds": dt_max,
                            "first_pass_seconds_rank0": cold,
                            "what": "plain FASTQ under %s: 2 files x %d reads per rank + one file of %d reads, ONE list scanned by "
                                    "all ranks through skh_scan_list (items dealt by size, the big file cut at checked record "
                                    "boundaries), counts all-reduced; decode threads per rank = SK_THREADS or the CPU budget / ranks; three timed "
                                    "passes behind an untimed first one (which page-locks the threads' buffers, kept by the context)"
                                    % (root, n_small, 4 * n_small)}
        barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)

    sd_leg = None
    if rank == 0 and world == 1 and not args.no_sd and not args.ablate:
        ctx.dev_free(dev)                                   # (the 1.5 GB stream is no longer needed)
        try:
            sd_leg = strain_detect_leg(device)
        except AssertionError:
            raise
        except Exception as e:                              # noqa: BLE001  (a side measurement must not take the headline line down)
            sd_leg = {"skipped": f"{type(e).__name__}: {e}"}

    if rank == 0:
        total_bases = nbases * args.steps * world
        value = total_bases / elapsed
        avg_ms = kern_ms / max(launches, 1)
        hits_rank0 = hits_per_pass if world == 1 else (facts["ranks"][0]["sum"] if facts else hits_per_pass // world)
        compulsory = float(nbytes) + HIT_BYTES * hits_rank0          # bytes one launch MUST move: stream once + counter RMW per hit
        achieved = compulsory / (avg_ms * 1e-3) / 1e9
        kernel_name = "sk_scan_grid"
        traffic, traffic_note = measured_traffic(args, kernel_name)
        line = {
            "metric": "metagenome bases/sec scanned at k=31", "value": value, "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_ms": args.preheat_ms, "preheat_launches": preheat_launches,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[1]: 5 Mbp synthetic strain (-r) vs %d x %d bp synthetic reads (-B) per GPU, k=31, "
                                   "reads resident in HBM as a record stream" % (args.reads, args.read_len),
                       "strain_keys": int(ks.nrows), "reads_per_gpu": args.reads, "read_len": args.read_len,
                       "strain_read_fraction": args.hit_frac,
                       "bases_per_step_per_gpu": nbases, "hits_per_pass_rank0_or_sum": hits_per_pass,
                       "pcie_inclusive_bases_per_s_host_buffers": host_rate,
                       "pcie_inclusive_bases_per_s_pinned_buffers": pinned_rate,
                       "packed_input_resident": packed_resident,
                       "file_fed_rank_sharded": file_fed,
                       "sharding": "reads sharded by rank, table replicated, one RCCL all-reduce of counts" if world > 1 else "single GPU"},
            "parity": parity,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kernel_name, "avg_launch_ms": avg_ms, "launches": int(launches),
                         "compulsory_bytes_per_launch": compulsory,
                         "compulsory": "record stream read once (%d B incl. separators) + %g B x %d hits" % (nbytes, HIT_BYTES, hits_rank0),
                         "survey_model_bytes_per_base": SURVEY_MODEL_BYTES_PER_BASE,
                         "survey_model_gbs": SURVEY_MODEL_BYTES_PER_BASE * nbases / (avg_ms * 1e-3) / 1e9,
                         "note": "frac = compulsory bytes / kernel time / 8 TB/s.  survey_model_* is SURVEY 8(d)'s 7.4 B/base "
                                 "(one 8 B probe per window), a model of a different algorithm kept for continuity only",
                         # What holds the kernel below the HBM roofline: the design asks ONE membership question
                         # per aligned 16-base chunk -- a random 8-byte load from the L2-resident filter -- and reads the stream in
                         # 128-byte requests; no exact scheme asks fewer (a 31-base window is guaranteed to hold only one aligned
                         # 16-mer, and the chunks of unrelated reads are independent random keys).  Requests per launch / kernel time
                         # against the L2's measured request ceiling is the kernel's second stated roofline.
                         "l2_requests": {"per_launch": nbytes / 16.0 + nbytes / 128.0,
                                         "what": "one level-1 filter block per 16-byte chunk + one request per 128 stream bytes (algorithmic; "
                                                 "measured TCC_REQ in profiles/)",
                                         "achieved_per_s": (nbytes / 16.0 + nbytes / 128.0) / (avg_ms * 1e-3), "ceiling_per_s": L2_REQ_CEILING,
                                         "frac": (nbytes / 16.0 + nbytes / 128.0) / (avg_ms * 1e-3) / L2_REQ_CEILING,
                                         "ceiling_source": "random 8-byte loads from a <= 4 MiB table, profiles/r01_gather_microbench.txt",
                                         # round 4 (tools/probes/overlap_probe.hip, profiles/r04_overlap_probe.txt; NOT measured in this run): the
                                         # kernel's two access patterns bare -- the stream's nt loads + one hashed 8-byte lookup per 16 bytes, no
                                         # arithmetic, same grid -- take 0.511 ms for this batch: what the memory system gives the single-kernel design
                                         "bare_access_pattern_ms": 0.511 * nbytes / 1.51e9,
                                         "frac_of_bare_access_pattern": (0.511 * nbytes / 1.51e9) / avg_ms if avg_ms > 0 else None},
                         # ... and the third: the SIMDs' issue slots (the kernel is integer SWAR work, 13 vector instructions per base)
                         "vector_issue": measured_vector_issue()},
            "cpu_baseline": cpu,
            "strain_detect": sd_leg,
        }
        if cpu is not None:
            # the north star's ">= 50 x the CPU reference" read against protocol (b) of SURVEY 8(d): P processes on the CPUs
            # this job is GRANTED (cpu_baseline.cores -- a cgroup share of the host, not a socket).  Resident = `value`;
            # file-fed = the product's list walk from plain FASTQ files (decode threads -> pinned buffers -> PCIe -> kernel).
            line["vs_cpu_baseline"] = {
                "resident": value / cpu["value"],
                "file_fed_plain_fastq": (file_fed["bases_per_s"] / cpu["value"]) if file_fed and "bases_per_s" in file_fed else None,
                "pcie_pinned": (pinned_rate / cpu["value"]) if pinned_rate else None,
                "what": "this run's rates / cpu_baseline.value (the unmodified reference as P = cpu_baseline.cores processes); "
                        ".gz end to end is bound by host inflate and is measured by tools/e2e_bench.py (DESIGN.md section 5)"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
