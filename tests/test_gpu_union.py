"""One table for several resident strains (sk_union_*): a batch tallied against the union must give, member by member,
exactly what the member's own table gives (sk_tally_batch) -- per-record tallies and the log of informative hits with
the member's own rows.  Strains that share most of their k-mers (diverged copies, exact copies, a repeated segment) are
the point: a shared key has ONE slot in the union."""
import os
import random

import numpy as np
import pytest

import _synth
import strainer2_amd as sk

pytestmark = pytest.mark.gpu


def _mutate(rng, seq, rate):
    b = bytearray(seq)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = rng.choice(b"ACGT")
    return bytes(b)


def _strains(rng, kind, n):
    base = _synth.rand_dna(rng, rng.choice([400, 3000, 40000]))
    out = []
    for s in range(n):
        pick = (s + kind) % 5
        if pick == 0:
            g = base                                           # the same strain again: every key shared
        elif pick == 1:
            g = _mutate(rng, base, rng.choice([0.002, 0.01, 0.05]))
        elif pick == 2:
            g = _synth.rand_dna(rng, rng.choice([200, 5000, 30000]))
        elif pick == 3:
            g = _synth.revcomp(base[len(base) // 3:]) + _synth.rand_dna(rng, 500)     # the other strand of a part of it
        else:
            cut = len(base) // 2
            g = base[:cut] + base[cut // 2:cut] + base[cut:]   # a repeated segment
            if len(g) > 100:
                g = g[:50] + b"N" + g[51:]
        out.append(g)
    return out


def _reads(rng, strains, nreads):
    recs = []
    for _ in range(nreads):
        ln = rng.choice([31, 32, 47, 64, 100, 150, 151, 250, 700])
        r = rng.random()
        if r < 0.7:
            g = strains[rng.randrange(len(strains))]
            if len(g) > ln:
                a = rng.randrange(len(g) - ln)
                seq = _mutate(rng, g[a:a + ln], rng.choice([0.0, 0.0, 0.01, 0.05]))
                if rng.random() < 0.5:
                    seq = _synth.revcomp(seq)
            else:
                seq = _synth.rand_dna(rng, ln)
        else:
            seq = _synth.rand_dna(rng, ln)
        if rng.random() < 0.05:
            seq = seq[:ln // 2] + b"N" + seq[ln // 2 + 1:]
        if rng.random() < 0.03:
            seq = seq.lower()
        recs.append(seq)
    return recs


def _check(seed, n, nreads=1500, hits_cap=None):
    rng = random.Random(seed)
    strains = _strains(rng, seed, n)
    recs = _reads(rng, strains, nreads)
    stream = b"\n".join(recs) + b"\n"
    starts = np.cumsum([0] + [len(r) + 1 for r in recs[:-1]]).astype(np.uint32)
    ctxs, sets, want = [], [], []
    try:
        for g in strains:
            ks = sk.Keyset.from_stream(g + b"\n", default_val=1, incr=0)
            c = sk.KmerContext(0)
            c.load_keyset(ks, 6)
            typ = np.ones(ks.nrows, dtype=np.uint32)
            if ks.nrows:
                typ[np.array(sorted(rng.sample(range(ks.nrows), max(1, ks.nrows // rng.choice([3, 10, 50])))))] = 2
            c.set_counts(0, typ)
            ctxs.append(c)
            sets.append(ks)
            want.append(c.tally_batch(stream, starts, 0, 2))
        with sk.KmerUnion(ctxs, 0, 2) as u:
            assert u.rows == sum(k.nrows for k in sets)
            tally, hits = u.tally_batch(stream, starts, hits_cap=hits_cap)
            tally2, hits2 = u.tally_batch(stream, starts)       # and once more: nothing is left over from a launch
        assert np.array_equal(tally, tally2) and np.array_equal(hits, hits2)
        for s in range(n):
            wt, wh = want[s]
            assert np.array_equal(tally[:, s, :], wt), (seed, s)
            mine = hits[hits[:, 0] == s][:, 1:]
            wh = wh[np.lexsort((wh[:, 1], wh[:, 0]))]
            assert np.array_equal(mine, wh), (seed, s, len(mine), len(wh))
        assert int(tally.sum()) > 0
    finally:
        for c in ctxs:
            c.close()
        for k in sets:
            k.close()


# SK_FUZZ_EXTRA=N: N more worlds on top of the twelve that always run (a longer hunt, run by hand), from SK_FUZZ_BASE on
EXTRA = int(os.environ.get("SK_FUZZ_EXTRA", "0"))
BASE = int(os.environ.get("SK_FUZZ_BASE", "1000"))


@pytest.mark.parametrize("seed", list(range(12)) + list(range(BASE, BASE + EXTRA)))
def test_union_equals_member_by_member(seed):
    _check(seed, 2 + seed % 6 if seed % 11 else 9 + seed % 23, nreads=1500 if seed < 12 else 300 + (seed * 37) % 2500)


def test_union_of_one_and_of_thirty_two():
    _check(100, 1, nreads=400)
    _check(101, 32, nreads=600)


def test_union_log_overflow_is_reported_and_recovered():
    _check(7, 4, hits_cap=5)


def test_union_refuses_what_it_cannot_hold():
    g = _synth.rand_dna(random.Random(3), 500)
    with sk.KmerContext(0) as a, sk.KmerContext(0) as b:
        ka = sk.Keyset.from_stream(g + b"\n", default_val=1, incr=0)
        kb = sk.Keyset.from_stream(g[:200] + b"R" + g[201:] + b"\n", default_val=1, incr=0)     # an IUPAC letter: byte-string keys
        a.load_keyset(ka, 6)
        b.load_keyset(kb, 6)
        with pytest.raises(sk.SKError):
            sk.KmerUnion([a, b])
        with pytest.raises(sk.SKError):
            sk.KmerUnion([])
        a.set_option("text_stage", 0)
        with pytest.raises(sk.SKError):
            sk.KmerUnion([a])


def test_union_a_million_short_records_against_the_members():
    """more than 16 x 65,536 records in one batch, most of them hitting: the compaction's workgroups take sixteen sets of 64 records
    each (sk_union_post), the pairs and log entries outnumber what travels home with the counters (SK_UNION_EAGER: the rest is
    fetched behind them) -- member by member the tallies and the log must be the member's own"""
    rng = np.random.default_rng(20260405)
    prng = random.Random(7)
    base = _synth.rand_dna(prng, 60000)
    strains = [base, _mutate(prng, base, 0.01), _synth.rand_dna(prng, 30000), _synth.revcomp(base[20000:]) + _synth.rand_dna(prng, 500)]
    nrec, L = 1_050_000, 48
    arrs = [np.frombuffer(g, dtype=np.uint8) for g in strains]
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(nrec, L + 1))].copy()
    which = rng.integers(0, len(strains) + 2, size=nrec)                  # (a third of the reads are random)
    for s, a in enumerate(arrs):
        rows = np.nonzero(which == s)[0]
        st = rng.integers(0, len(a) - L, size=len(rows))
        reads[rows, :L] = a[st[:, None] + np.arange(L)[None, :]]
    reads[:, L] = ord("\n")
    stream = reads.tobytes()
    starts = (np.arange(nrec, dtype=np.uint64) * (L + 1)).astype(np.uint32)
    ctxs, sets, want = [], [], []
    try:
        for g in strains:
            ks = sk.Keyset.from_stream(g + b"\n", default_val=1, incr=0)
            c = sk.KmerContext(0)
            c.load_keyset(ks, 6)
            typ = np.ones(ks.nrows, dtype=np.uint32)
            typ[::37] = 2
            c.set_counts(0, typ)
            ctxs.append(c)
            sets.append(ks)
            want.append(c.tally_batch(stream, starts, 0, 2))
        with sk.KmerUnion(ctxs, 0, 2) as u:
            tally, hits = u.tally_batch(stream, starts, hits_cap=8_000_000)
        assert int((tally[:, :, 0] > 0).sum()) > 16384 * 8 and len(hits) > 16384 * 4       # (far beyond what comes back with the counters)
        for s in range(len(strains)):
            wt, wh = want[s]
            assert np.array_equal(tally[:, s, :], wt), s
            mine = hits[hits[:, 0] == s][:, 1:]
            wh = wh[np.lexsort((wh[:, 1], wh[:, 0]))]
            assert np.array_equal(mine, wh), (s, len(mine), len(wh))
    finally:
        for c in ctxs:
            c.close()
        for k in sets:
            k.close()
