#!/usr/bin/env python3
"""Differential hunt for `strain_detect -S` (many strains in one pass: union table, sparse per-strain replay, chunk upload
ahead, several parser threads) against the oracle program run strain by strain.  Per seed: 2-6 RELATED strains (diverged and
exact copies of ancestors, the other strand), informative lists that overlap only in part, a -B list with SE / PE / PEI files
(plain and .gz, FASTA and FASTQ), a third of the reads cut below k (they re-emit the tallies of the read before), N's, tiny
chunks (SK_SD_CHUNK_BYTES) so that mates, short-read runs and hits straddle chunk borders, several parser threads; every third seed with the
strains dealt to several logical devices (SK_DEVICES), every fifth without the scan-ahead of the next chunk.
SEEDS=a..b (default 0..19).  Test tool: prints one line per seed and exits non-zero on the first difference."""
import gzip
import os
import random
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(REPO, "strainer2_amd", "bin", "strain_detect")
ORA = os.path.join(REPO, "oracle", "ksd_oracle")
COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def dna(r, n):
    return bytes(r.choice(b"ACGT") for _ in range(n))


def mutate(r, s, rate):
    b = bytearray(s)
    for i in range(len(b)):
        if r.random() < rate:
            b[i] = r.choice(b"ACGT")
    return bytes(b)


def canon(k):
    rc = k.translate(COMP)[::-1]
    return k if k >= rc else rc


def one_seed(seed, d):
    r = random.Random(seed)
    ns = r.randint(2, 6)
    anc = [dna(r, r.choice([3000, 9000, 20000])) for _ in range(2)]
    strains = []
    for s in range(ns):
        g = mutate(r, anc[s % 2], r.choice([0.0, 0.004, 0.02]))
        if r.random() < 0.3:
            g = g.translate(COMP)[::-1]
        if r.random() < 0.3:
            g = g[:len(g) // 2] + b"N" + g[len(g) // 2:]
        strains.append(g)
        w = r.choice([60, 70, 0])
        body = b"\n".join(g[i:i + w] for i in range(0, len(g), w)) if w else g
        open(f"{d}/s{s}.fa", "wb").write(b">s%d x\n" % s + body + b"\n")
        step = r.choice([3, 7, 19])
        kms = sorted({canon(g[i:i + 31]) for i in range(s % step, len(g) - 31, step) if b"N" not in g[i:i + 31]})
        open(f"{d}/s{s}.inf", "wb").write(b"#inf\n" + b"\n".join(kms) + b"\n")

    def reads(n):
        out = []
        for _ in range(n):
            L = r.choice([150, 150, 100, 250, 31, 32])
            if r.random() < 0.45:
                g = strains[r.randrange(ns)]
                a = r.randrange(0, max(1, len(g) - L))
                rd = mutate(r, g[a:a + L], r.choice([0.0, 0.0, 0.01]))
                if r.random() < 0.5:
                    rd = rd.translate(COMP)[::-1]
            else:
                rd = dna(r, L)
            if r.random() < 0.3:
                rd = rd[:r.randrange(0, 31)]
            if r.random() < 0.03 and len(rd) > 10:
                rd = rd[:5] + b"N" + rd[6:]
            out.append(rd)
        return out

    def write(path, rs, fq, gz):
        if fq:
            body = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, x, b"I" * len(x)) for i, x in enumerate(rs))
        else:
            body = b"".join(b">r%d c\n%s\n" % (i, x) for i, x in enumerate(rs))
        (gzip.open(path, "wb", compresslevel=1) if gz else open(path, "wb")).write(body)

    n = r.choice([300, 1500, 4000])
    lines = []
    for k in range(r.randint(1, 3)):
        mode = r.choice(["SE", "PE", "PEI"])
        fq, gz = r.random() < 0.5, r.random() < 0.4
        ext = (".fq" if fq else ".fa") + (".gz" if gz else "")
        if mode == "PE":
            na, nb = n, n - r.choice([0, 0, 1, 7])            # (a mate file that ends first)
            write(f"{d}/m{k}_1{ext}", reads(na), fq, gz)
            write(f"{d}/m{k}_2{ext}", reads(nb), fq, gz)
            lines.append(f"PE\t{d}/m{k}_1{ext}\t{d}/m{k}_2{ext}\n")
        else:
            write(f"{d}/m{k}{ext}", reads(n + (mode == "PEI" and r.random() < 0.5)), fq, gz)
            lines.append(f"{mode}\t{d}/m{k}{ext}\n")
    open(f"{d}/B.txt", "w").write("".join(lines))
    open(f"{d}/strains.txt", "w").write("".join(f"{d}/s{s}.fa\t{d}/s{s}.inf\t{d}/multi{s}.gz\n" for s in range(ns)))
    env = dict(os.environ, SK_SD_CHUNK_BYTES=str(r.choice([700, 5000, 60000, 33554432])), SK_PARSE_THREADS=str(r.choice([1, 2, 4])),
               SK_GZ_THREADS=str(r.choice([1, 3])), SK_GZ_SEG="4000")
    if r.random() < 0.25:
        env["SK_SD_NO_UNION"] = "1"
    if seed % 3 == 1:                                   # one process, several (logical) devices on the one card: groups of 1-2 strains dealt to them
        env["SK_DEVICES"] = r.choice(["0,0", "0,0,0"])
        env["SK_SD_GROUP"] = str(r.choice([1, 2]))
    if seed % 5 == 2:
        env["SK_SD_NO_AHEAD"] = "1"
    p = subprocess.run([EXE, "-S", f"{d}/strains.txt", "-B", f"{d}/B.txt"], capture_output=True, env=env)

    def ora(s):
        return subprocess.run([ORA, "-r", f"{d}/s{s}.fa", "-a", f"{d}/s{s}.inf", "-B", f"{d}/B.txt", "-o", f"{d}/ora{s}.gz"], capture_output=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        os_ = list(ex.map(ora, range(ns)))
    lines_out = 0
    for s in range(ns):
        want_rc = os_[s].returncode
        if want_rc != 0 or p.returncode != 0:
            if (p.returncode != 0) != (want_rc != 0):
                return f"seed {seed}: exit status {p.returncode} vs oracle {want_rc} (strain {s}): {p.stderr[-300:]!r} / {os_[s].stderr[-300:]!r}"
            continue
        a, b = gzip.open(f"{d}/multi{s}.gz").read(), gzip.open(f"{d}/ora{s}.gz").read()
        if a != b:
            return f"seed {seed}: strain {s} of {ns} differs ({a.count(bytes([10]))} vs {b.count(bytes([10]))} lines); env {dict((k, v) for k, v in env.items() if k.startswith('SK_'))}"
        lines_out += a.count(b"\n")
    print(f"seed {seed}: {ns} strains, {len(lines)} list lines, {lines_out} output lines: identical", flush=True)
    return None


lo, hi = (int(x) for x in os.environ.get("SEEDS", "0..19").split(".."))
for seed in range(lo, hi + 1):
    d = tempfile.mkdtemp(prefix="sk_sdmd_")
    try:
        err = one_seed(seed, d)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if err:
        print(err)
        sys.exit(1)
print("all identical")
