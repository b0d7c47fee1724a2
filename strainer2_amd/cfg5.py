"""One GPU's share of BASELINE configs[4] as files (SURVEY 8(d) recipe): `strain_detect` with 32 strains resident (256 strains / 8 GPUs)
against one 100 Gbase SE metagenome.

  strains   NSTRAINS = 32 genomes of 5 Mbp (one record each, i.i.d. uniform ACGT, seed SEED_S + s)
  -a lists  per strain the 31-mers that start at 1 % of its positions (seed SEED_I + s), one per line, as the strain spells them
            (src/strain_detect.c:668-726 orients every line itself)
  reads     READS = 66,666,667 reads x 150 bp of FASTA (10.0 Gbase, 154 bytes per record): 2 % of the reads are cut from one of the
            32 strains (0.5 % substitutions, half of them reverse-complemented), the rest i.i.d. uniform ACGT, 0.01 % N; written in
            blocks of BLOCK reads, block b from seed SEED_R + b -- any process can write any block
  -B list   the reads file listed LIST_REPEAT = 10 times as `SE <path>`: 100 Gbase scanned
  prefix    the first PREFIX_READS = 6,666,667 reads (1.0 Gbase) as a file of their own: the part on which the unmodified reference
            program's output for two strains is pinned (tests/golden/make_cfg5_share_facts.py -> tests/golden/cfg5_share_facts.json)

Both sides (build container: reference; GPU box: tools/sd_cfg5_share.py) write the same bytes from these seeds.
"""
import os

import numpy as np

from . import synth

NSTRAINS = 32
STRAIN_BP = 5_000_000
READS = 66_666_667
READ_LEN = 150
BLOCK = 1_000_000
PREFIX_READS = 6_666_667
LIST_REPEAT = 10
PINNED_STRAINS = (0, 17)
SEED_S = synth.SEED + 20000
SEED_I = synth.SEED + 21000
SEED_R = synth.SEED + 22000
REC = 3 + READ_LEN + 1                                    # ">r\n" + bases + "\n"


def strain(s):
    return synth._rand_bases(np.random.default_rng(SEED_S + s), STRAIN_BP)


def informative_lines(s, g):
    rng = np.random.default_rng(SEED_I + s)
    pos = np.sort(rng.choice(g.size - 30, size=(g.size - 30) // 100, replace=False))
    win = np.lib.stride_tricks.sliding_window_view(g, 31)[pos]
    rows = np.empty((pos.size, 32), dtype=np.uint8)
    rows[:, :31] = win
    rows[:, 31] = 10
    return rows.tobytes()


def write_strain(d, s):
    g = strain(s)
    with open(os.path.join(d, f"s{s}.fa"), "wb") as f:
        f.write(b">s%d\n" % s + g.tobytes() + b"\n")
    with open(os.path.join(d, f"s{s}.inf"), "wb") as f:
        f.write(informative_lines(s, g))


def reads_block(b, m, genome):
    """FASTA bytes of the first m reads of block b; genome = the 32 strains end to end.  A block is always DRAWN whole (BLOCK
    reads) and then cut: what read i of block b is must not depend on how many reads of the block are wanted (the prefix file
    ends inside a block)."""
    return _whole_block(b, genome)[:m]


def _whole_block(b, genome):
    m = BLOCK
    rng = np.random.default_rng(SEED_R + b)
    blk = synth._rand_bases(rng, m * READ_LEN).reshape(m, READ_LEN)
    h = np.flatnonzero(rng.random(m) < 0.02)
    if h.size:
        si = rng.integers(0, NSTRAINS, size=h.size)
        start = (rng.random(h.size) * (STRAIN_BP - READ_LEN + 1)).astype(np.int64)
        idx = (si.astype(np.int64) * STRAIN_BP + start)[:, None] + np.arange(READ_LEN)[None, :]
        sub = genome[idx]
        flip = rng.random(h.size) < 0.5
        sub[flip] = synth._COMP[sub[flip]][:, ::-1]
        mut = rng.random(sub.shape) < 0.005
        sub[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
        blk[h] = sub
    k = rng.binomial(m * READ_LEN, 1e-4)
    if k:
        blk.reshape(-1)[rng.integers(0, m * READ_LEN, size=k)] = ord("N")
    fa = np.empty((m, REC), dtype=np.uint8)
    fa[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
    fa[:, 3:3 + READ_LEN] = blk
    fa[:, 3 + READ_LEN] = 10
    return fa


_GENOME = None


def _job(args):
    global _GENOME
    kind, d, i, reads = args
    if kind == "s":
        write_strain(d, i)
        return kind, i
    if _GENOME is None:
        _GENOME = np.concatenate([strain(s) for s in range(NSTRAINS)])
    m = min(BLOCK, reads - i * BLOCK)
    fa = reads_block(i, m, _GENOME)
    fd = os.open(os.path.join(d, "reads.fa"), os.O_WRONLY)
    try:
        os.pwrite(fd, fa.tobytes(), i * BLOCK * REC)
    finally:
        os.close(fd)
    return kind, i


def write_all(d, procs=8, reads=READS, prefix_reads=PREFIX_READS, only_prefix=False, progress=None):
    """strains, -a lists, the -S list, the reads (or only their first prefix_reads), the prefix file, the -B list; forks workers:
    call before the process touches the GPU.  Returns the paths."""
    import multiprocessing as mp
    os.makedirs(d, exist_ok=True)
    n_reads = prefix_reads if only_prefix else reads
    nblocks = (n_reads + BLOCK - 1) // BLOCK
    with open(os.path.join(d, "reads.fa"), "wb") as f:
        f.truncate(n_reads * REC)
    jobs = [("s", d, s, None) for s in range(NSTRAINS)] + [("r", d, b, n_reads) for b in range(nblocks)]
    with mp.get_context("fork").Pool(procs) as pool:
        for n, _ in enumerate(pool.imap_unordered(_job, jobs)):
            if progress and n % 16 == 0:
                progress(n, len(jobs))
    with open(os.path.join(d, "reads.fa"), "rb") as src, open(os.path.join(d, "prefix.fa"), "wb") as dst:
        left = prefix_reads * REC
        while left:
            blk = src.read(min(left, 1 << 26))
            dst.write(blk)
            left -= len(blk)
    with open(os.path.join(d, "strains.txt"), "w") as f:
        f.write("".join(f"s{s}.fa\ts{s}.inf\tmulti{s}.gz\n" for s in range(NSTRAINS)))
    with open(os.path.join(d, "strains_prefix.txt"), "w") as f:
        f.write("".join(f"s{s}.fa\ts{s}.inf\tprefix{s}.gz\n" for s in range(NSTRAINS)))
    with open(os.path.join(d, "B.txt"), "w") as f:
        f.write("SE\treads.fa\n" * LIST_REPEAT)
    # (every path is relative: the programs run with cwd = d, and the hit lines, which name the reads file, read the same everywhere)
    return {"strains": "strains.txt", "strains_prefix": "strains_prefix.txt", "B": "B.txt", "reads": "reads.fa", "prefix": "prefix.fa"}
