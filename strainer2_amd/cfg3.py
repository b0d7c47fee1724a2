"""BASELINE configs[2] AT SPEC as files (SURVEY 8(d) recipe): one 5 Mbp strain (-r) against

  -A  1000 genomes x 5 Mbp (FASTA, 60-column lines, 50 contigs each): genomes 0..9 are the strain with 1 % substitutions
      (odd ones on the other strand, contig by contig), the others i.i.d. uniform ACGT; every 100th file is gzip'ed
  -B  N_B files x 1,000,000 reads x 150 bp of FASTQ (the cfg-2 read recipe: 2 % cut from the strain with 0.5 % substitutions,
      half of those reverse-complemented, 0.01 % N), each listed LIST_REPEAT = 10 times: 67 files = 10.05 Gbase distinct,
      100.5 Gbase scanned
  -C  5 genomes: random, THE -r PATH ITSELF (the skip rule, src/genome_compare.c:138-141), the strain with 2 % substitutions,
      two more random ones
  -p  a progress file

Everything is a function of fixed seeds (numpy PCG64), file by file, so that the build container (where the unmodified reference
program produces the expected facts, tests/golden/make_cfg3_full_facts.py) and the GPU box (tools/cfg3_full.py) write the same
bytes independently and in parallel.  Nothing but the facts is committed.
"""
import gzip
import os

import numpy as np

from . import synth

N_GENOMES = 1000
N_B_FILES = 67
READS_PER_FILE = 1_000_000
READ_LEN = 150
LIST_REPEAT = 10
SEED_A = synth.SEED + 1000
SEED_B = synth.SEED + 5000
SEED_C = synth.SEED + 9000


def fasta_bytes(contigs, tag, width=60):
    """FASTA text of a list of equally long uint8 contigs, `width` bases per line (vectorised)"""
    out = []
    for i, c in enumerate(contigs):
        out.append(b">%s_%d\n" % (tag, i + 1))
        full = (c.size // width) * width
        if full:
            rows = np.empty((full // width, width + 1), dtype=np.uint8)
            rows[:, :width] = c[:full].reshape(-1, width)
            rows[:, width] = 10
            out.append(rows.tobytes())
        if c.size > full:
            out.append(c[full:].tobytes() + b"\n")
    return b"".join(out)


def _diverged(contigs, rate, seed, flip):
    """the strain's contigs with `rate` substitutions (always to a different base; N stays), optionally reverse-complemented"""
    rng = np.random.default_rng(seed)
    out = []
    for c in contigs:
        d = c.copy()
        m = (rng.random(d.size) < rate) & (d != ord("N"))
        code = (np.searchsorted(synth._ACGT, d[m]) + rng.integers(1, 4, int(m.sum()))) & 3       # _ACGT is sorted: A C G T
        d[m] = synth._ACGT[code]
        out.append(synth._COMP[d][::-1].copy() if flip else d)
    return out


def genome_contigs(i, strain):
    if i < 10:
        return _diverged(strain, 0.01, SEED_A + i, flip=bool(i & 1))
    return synth.make_strain(n_enn=0, seed=SEED_A + i)


def genome_name(i):
    return "genomes/g%04d.fa%s" % (i, ".gz" if i % 100 == 50 else "")


def write_genome(d, i, strain):
    data = fasta_bytes(genome_contigs(i, strain), b"g%d" % i)
    p = os.path.join(d, genome_name(i))
    if p.endswith(".gz"):
        with gzip.GzipFile(p, "wb", compresslevel=1, mtime=0) as f:
            f.write(data)
    else:
        with open(p, "wb") as f:
            f.write(data)


def reads_name(j):
    return "reads/mg%03d.fq" % j


def write_reads(d, j, strain, n_reads=READS_PER_FILE):
    reads, _ = synth.make_reads(strain, n_reads, READ_LEN, hit_frac=0.02, seed=SEED_B + j)
    rec = READ_LEN + 1
    rows = reads.reshape(n_reads, rec)
    fq = np.empty((n_reads, 3 + rec + 2 + rec), dtype=np.uint8)
    fq[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    fq[:, 3:3 + rec] = rows
    fq[:, 3 + rec:5 + rec] = np.frombuffer(b"+\n", dtype=np.uint8)
    fq[:, 5 + rec:5 + rec + READ_LEN] = ord("F")
    fq[:, 5 + rec + READ_LEN] = 10
    fq.tofile(os.path.join(d, reads_name(j)))


C_NAMES = ["genomes/drug0.fa", "strain.fa", "genomes/drug2.fa", "genomes/drug3.fa", "genomes/drug4.fa"]


def write_drug(d, k, strain):
    if k == 1:
        return                                            # the -r path itself
    contigs = _diverged(strain, 0.02, SEED_C + k, flip=False) if k == 2 else synth.make_strain(n_enn=0, seed=SEED_C + k)
    with open(os.path.join(d, C_NAMES[k]), "wb") as f:
        f.write(fasta_bytes(contigs, b"drug%d" % k, width=80))


def write_lists(d, n_genomes=N_GENOMES, n_b=N_B_FILES, repeat=LIST_REPEAT):
    with open(os.path.join(d, "A.txt"), "w") as f:
        f.write("".join(genome_name(i) + "\n" for i in range(n_genomes)))
    with open(os.path.join(d, "B.txt"), "w") as f:       # the whole set, `repeat` times over (a list line is scanned as often as it is listed)
        f.write("".join(reads_name(j) + "\n" for _ in range(repeat) for j in range(n_b)))
    with open(os.path.join(d, "C.txt"), "w") as f:
        f.write("".join(n + "\n" for n in C_NAMES))
    return ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-C", "C.txt", "-p", "progress.txt"]


def _job(args):
    kind, d, idx, extra = args
    strain = synth.make_strain()
    if kind == "g":
        write_genome(d, idx, strain)
    elif kind == "b":
        write_reads(d, idx, strain, extra)
    else:
        write_drug(d, idx, strain)
    return kind, idx


def write_all(d, procs=8, n_genomes=N_GENOMES, n_b=N_B_FILES, reads_per_file=READS_PER_FILE, repeat=LIST_REPEAT, progress=None):
    """every input under directory d (relative paths inside the lists, as in the bundled example); returns the argv.
    Must run before the process touches the GPU (it forks workers)."""
    import multiprocessing as mp
    os.makedirs(os.path.join(d, "genomes"), exist_ok=True)
    os.makedirs(os.path.join(d, "reads"), exist_ok=True)
    strain = synth.make_strain()
    with open(os.path.join(d, "strain.fa"), "wb") as f:
        f.write(synth.strain_fasta(strain))
    jobs = [("b", d, j, reads_per_file) for j in range(n_b)] + [("g", d, i, None) for i in range(n_genomes)] + [("c", d, k, None) for k in range(5)]
    with mp.get_context("fork").Pool(procs) as pool:
        for n, _ in enumerate(pool.imap_unordered(_job, jobs, chunksize=4)):
            if progress and n % 100 == 0:
                progress(n, len(jobs))
    return write_lists(d, n_genomes, n_b, repeat)
