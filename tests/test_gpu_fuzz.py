"""Differential fuzzing of the scan against the oracle: many small random worlds (strain size and content,
IUPAC letters and U in strain and reads, low-complexity and repeated stretches, duplicated reads, read lengths
around k and around the 16-byte chunk grid, junk rates, streams that cross tile boundaries), all three scanned
columns compared count for count, then the same through TALLY mode (per-read tallies and the hit log)."""
import os
import random

import numpy as np
import pytest

import _oracle
import _synth
import strainer2_amd as sk

pytestmark = pytest.mark.gpu

# SK_FUZZ_EXTRA=N: N more seeds per test on top of the ones that always run (a longer hunt, run by hand)
EXTRA = int(os.environ.get("SK_FUZZ_EXTRA", "0"))
BASE = int(os.environ.get("SK_FUZZ_BASE", "1000"))        # first seed of the extra worlds


def world(seed):
    rng = random.Random(seed)
    kind = seed % 6
    n = rng.choice([40, 300, 3000, 30000, 120000])
    strain = bytearray(_synth.rand_dna(rng, n))
    if kind == 1:                                      # low complexity and tandem repeats
        unit = _synth.rand_dna(rng, rng.choice([1, 2, 3, 7, 16, 31, 33]))
        at = rng.randrange(0, max(1, n - 400))
        rep = (unit * 400)[:min(400, n - at)]
        strain[at:at + len(rep)] = rep
    if kind == 2:                                      # IUPAC letters, U and N in the strain
        for ch in b"RYKMSWBDHVUNn":
            strain[rng.randrange(n)] = ch
    if kind == 3 and n > 200:                          # the strain repeats itself (multiplicities > 1)
        strain[n // 2:n // 2 + 100] = strain[10:110]
    cut = rng.randrange(1, n)
    sstream = bytes(strain[:cut]) + b"\n" + (bytes(strain[cut:]).lower() if kind == 4 else bytes(strain[cut:])) + b"\n"
    junk = [0.0, 0.001, 0.01, 0.05][seed % 4]
    lens = [(0, 64), (28, 34), (0, 200), (140, 160), (300, 2000), (15, 49)][seed % 6]
    nreads = max(50, min(4000, 400_000 // (lens[1] + 1)))
    data = _synth.fuzz_stream(rng, bytes(strain), nreads, p_junk=junk, min_len=lens[0], max_len=lens[1])
    if kind == 5:                                      # the same read over and over
        one = bytes(strain[:min(n, 150)]).replace(b"N", b"A") + b"\n"
        data += one * 3000
    return sstream, data


@pytest.mark.parametrize("seed", list(range(36)) + list(range(BASE, BASE + EXTRA)))
def test_scan_count_fuzz(seed):
    sstream, data = world(seed)
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream, short_policy=1) == 0          # (records under k-1 bases: skipped, as the product does)
    with sk.KmerContext(0) as c:
        if seed % 5 == 0:
            c.set_option("text_stage", 0)              # every window probed on its own now and then
        c.set_option("pipeline", 2 if seed % 2 else 1)  # the partitioned pipeline on the odd seeds, the single kernel on the even ones
        c.load_keyset(ks, 4)
        third = len(data) // 3
        cuts = [0, data.rfind(b"\n", 0, third) + 1, data.rfind(b"\n", 0, 2 * third) + 1, len(data)]
        for col in (1, 2, 3):
            piece = data[cuts[col - 1]:cuts[col]]
            c.scan_stream(piece, col)
            t.scan_stream(piece, col)
        c.scan_stream(data, 2)                         # and the whole thing once more into one column
        t.scan_stream(data, 2)
        okeys, ocounts = t.rows()
        assert ks.keys() == okeys
        for col in range(4):
            assert np.array_equal(c.counts(col), ocounts[:, col]), (seed, col)


_CLEAN = bytes(b if b in b"ACGTacgtNn\n" else ord("N") for b in range(256))


@pytest.mark.parametrize("seed", list(range(200, 224)) + list(range(7 * BASE + 2000, 7 * BASE + 2000 + EXTRA // 3)))
def test_scan_count_fuzz_packed(seed):
    """the same worlds with every byte that is neither A/C/G/T, N/n nor a newline turned into N (a packed batch holds no other), scanned
    in the host-PACKED form (sk_pack_stream -> sk_scan_pinned_packed, chunks of whole records in pinned memory, ragged sizes) against the
    oracle's counts of the same bytes"""
    sstream, data = world(seed)
    data = data.translate(_CLEAN)
    ks = sk.Keyset.from_stream(sstream)
    t = _oracle.OracleTable()
    assert t.build_stream(sstream, short_policy=1) == 0
    rng = random.Random(seed)
    with sk.KmerContext(0) as c:
        if seed % 5 == 0:
            c.set_option("text_stage", 0)
        c.load_keyset(ks, 4)
        cap = rng.choice([1 << 13, 1 << 16, 1 << 20])
        recs = data.split(b"\n")[:-1]
        raw = np.empty(cap, dtype=np.uint8)
        pk = [c.pinned_alloc(6 * ((cap + 15) // 16)) for _ in range(2)]
        tickets = [None, None]
        i = k = 0
        while i < len(recs):
            n, j = 0, i
            while j < len(recs) and n + len(recs[j]) + 1 <= cap:
                n += len(recs[j]) + 1
                j += 1
            if j == i:                                  # (a record longer than the chunk: as bytes, the way the programs cut it)
                c.scan_stream(recs[i] + b"\n", 1)
                i += 1
                continue
            raw[:n] = np.frombuffer(b"\n".join(recs[i:j]) + b"\n", dtype=np.uint8)
            if tickets[k] is not None:
                c.ticket_wait(tickets[k])
            _, odd = sk.pack_stream(raw[:n], out=pk[k])
            assert not odd
            tickets[k] = c.scan_pinned_packed(pk[k], n, 1)
            k ^= 1
            i = j
        c.sync()
        t.scan_stream(data, 1)
        okeys, ocounts = t.rows()
        assert ks.keys() == okeys
        assert np.array_equal(c.counts(1), ocounts[:, 1]), seed
        for a in pk:
            c.pinned_free(a)


@pytest.mark.parametrize("seed", list(range(100, 112)) + list(range(4 * BASE + 1000, 4 * BASE + 1000 + EXTRA // 3)))
def test_tally_fuzz(seed):
    """per-read tallies and the log of informative hits against counts derived from the oracle's scan of each
    read on its own"""
    sstream, data = world(seed)
    ks = sk.Keyset.from_stream(sstream, default_val=1, incr=0)
    reads = [r for r in data.split(b"\n")[:-1] if len(r) >= 31][:400]
    if not reads or ks.nrows == 0:
        pytest.skip("no read of k bases in this world")
    rng = random.Random(seed)
    inf_rows = np.array(sorted(rng.sample(range(ks.nrows), max(1, ks.nrows // 7))))
    stream = b"\n".join(reads) + b"\n"
    starts = np.cumsum([0] + [len(r) + 1 for r in reads[:-1]])
    with sk.KmerContext(0) as c:
        c.set_option("pipeline", 2 if seed % 2 else 1)
        c.load_keyset(ks, 6)
        typ = np.ones(ks.nrows, dtype=np.uint32)
        typ[inf_rows] = 2
        c.set_counts(0, typ)
        tally, hits = c.tally_batch(stream, starts, 0, 2)
    t = _oracle.OracleTable(ncols=6)
    assert t.build_stream(sstream, default=1, incr=0, short_policy=1) == 0
    assert ks.keys() == t.rows()[0]
    is_inf = np.zeros(ks.nrows, dtype=bool)
    is_inf[inf_rows] = True
    prev = np.zeros(ks.nrows, dtype=np.int64)
    for i, r in enumerate(reads):                      # the oracle one read at a time: differences of its counters
        t.scan_stream(r + b"\n", 1)
        _, oc = t.rows()
        now = oc[:, 1].astype(np.int64)
        d = now - prev
        prev = now
        assert int(tally[i, 0]) == int(d.sum()), (seed, i)
        assert int(tally[i, 1]) == int(d[is_inf].sum()), (seed, i)
    assert len(hits) == int(tally[:, 1].sum())
