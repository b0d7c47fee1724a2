"""Fixed-seed synthetic workloads of SURVEY.md 8(d) (numpy PCG64, seed 0x5EED31).

strain : 5,000,000 bp as 50 contigs of 100 kbp, i.i.d. uniform ACGT, 10 isolated N.
reads  : n reads x 150 bp; a fraction `hit_frac` (0.02) are substrings of the strain (half of
         them reverse-complemented, 0.5 % substitutions), the rest i.i.d. uniform ACGT;
         0.01 % of all bases set to N.  Returned as the device ABI's record stream:
         sequence bytes, every record followed by '\\n'.
"""
import numpy as np

SEED = 0x5EED31
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[list(b"ACGTN")] = list(b"TGCAN")
_BYTE2BASE = bytes(b"ACGT"[i & 3] for i in range(256))


def _rand_bases(rng, n):
    """n i.i.d. uniform ACGT bytes (writable array)."""
    raw = rng.integers(0, 1 << 63, size=(n + 7) // 8, dtype=np.int64).tobytes()[:n]
    return np.frombuffer(bytearray(raw.translate(_BYTE2BASE)), dtype=np.uint8)


def make_strain(total_bp=5_000_000, contig_bp=100_000, n_enn=10, seed=SEED):
    """Returns (contigs: list[np.ndarray uint8], fasta_bytes)."""
    rng = np.random.default_rng(seed)
    seq = _rand_bases(rng, total_bp)
    if n_enn:
        seq[rng.choice(total_bp, size=n_enn, replace=False)] = ord("N")
    contigs = [seq[i:i + contig_bp] for i in range(0, total_bp, contig_bp)]
    return contigs


def strain_fasta(contigs, width=60):
    out = []
    for i, c in enumerate(contigs):
        out.append(b">contig%d\n" % (i + 1))
        b = c.tobytes()
        out.append(b"\n".join(b[j:j + width] for j in range(0, len(b), width)) + b"\n")
    return b"".join(out)


def strain_stream(contigs):
    """The strain as a record stream (one record per contig)."""
    return b"\n".join(c.tobytes() for c in contigs) + b"\n"


def make_reads(contigs, n_reads, read_len=150, hit_frac=0.02, sub_rate=0.005, enn_rate=1e-4, seed=SEED + 1,
               chunk=100_000):
    """Record stream (np.uint8, n_reads * (read_len + 1) bytes) and the number of bases."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    out[:, read_len] = ord("\n")
    clen = len(contigs[0])
    genome = np.concatenate(contigs)
    usable = [c for c in range(len(contigs)) if len(contigs[c]) >= read_len]
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        m = b - a
        blk = _rand_bases(rng, m * read_len).reshape(m, read_len)
        is_hit = rng.random(m) < hit_frac
        h = np.nonzero(is_hit)[0]
        if h.size:
            ci = rng.choice(np.asarray(usable), size=h.size)
            lens = np.asarray([len(contigs[c]) for c in ci])
            start = (rng.random(h.size) * (lens - read_len + 1)).astype(np.int64)
            idx = (ci.astype(np.int64) * clen + start)[:, None] + np.arange(read_len)[None, :]
            sub = genome[idx]
            flip = rng.random(h.size) < 0.5
            sub[flip] = _COMP[sub[flip]][:, ::-1]
            mut = rng.random(sub.shape) < sub_rate
            sub[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
            blk[h] = sub
        if enn_rate:
            k = rng.binomial(m * read_len, enn_rate)
            if k:
                pos = rng.integers(0, m * read_len, size=k)
                blk.reshape(-1)[pos] = ord("N")
        out[a:b, :read_len] = blk
    return out.reshape(-1), n_reads * read_len
