#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/cases/ and tests/golden/bundled/.

Run HERE (the build container), never on the GPU box:

    make -C oracle ref          # builds the UNMODIFIED reference into oracle/_ref/
    python tests/golden/make_golden.py

Every expected output below is produced by the real reference binary
(oracle/_ref/kmer_scrub_count, compiled from /root/reference/src in place).  The fixtures
are DATA: small synthetic inputs written by this script (fixed seed) and the bytes the
reference printed for them.  For the reference's bundled example (test/example.sh step 1)
the five input data files are copied as data and only md5 / line-count facts of the
254 MB output are recorded.
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(REPO, "oracle", "_ref", "kmer_scrub_count")
REF_SD = os.path.join(REPO, "oracle", "_ref", "strain_detect")
SD_CASES = os.path.join(HERE, "sd_cases")
REF_TEST = "/root/reference/test"
CASES = os.path.join(HERE, "cases")
BUNDLED = os.path.join(HERE, "bundled")

COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def rand_dna(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def wrap(s, w):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w))


def write(path, text, gz=False, binary=False):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = text if binary else text.encode("latin-1")
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def run_case(name, argv, note):
    d = os.path.join(CASES, name)
    p = subprocess.run([REF_BIN] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(os.path.join(d, "expected.stdout"), "wb") as f:
        f.write(p.stdout)
    with open(os.path.join(d, "expected.stderr"), "wb") as f:
        f.write(p.stderr)
    prog = None
    if "-p" in argv:
        pf = os.path.join(d, argv[argv.index("-p") + 1])
        if os.path.exists(pf):
            with open(pf) as f:
                prog = [ln.split("\t")[0].rstrip("\n") for ln in f]
            os.remove(pf)
    meta = {"argv": argv, "returncode": p.returncode, "note": note, "progress_col1": prog}
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"{name}: rc={p.returncode} stdout={len(p.stdout)}B stderr={len(p.stderr)}B")


def run_sd_case(name, argv, note):
    """strain_detect golden: stdout, stderr, status and the DECOMPRESSED -o file."""
    d = os.path.join(SD_CASES, name)
    p = subprocess.run([REF_SD] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(os.path.join(d, "expected.stdout"), "wb") as f:
        f.write(p.stdout)
    with open(os.path.join(d, "expected.stderr"), "wb") as f:
        f.write(p.stderr)
    hits = None
    if "-o" in argv:
        of = os.path.join(d, argv[argv.index("-o") + 1])
        if os.path.exists(of):
            with gzip.open(of, "rb") as f:
                hits = f.read()
            os.remove(of)
            with open(os.path.join(d, "expected.hits"), "wb") as f:
                f.write(hits)
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump({"argv": argv, "returncode": p.returncode, "note": note, "has_hits": hits is not None}, f, indent=1)
    print(f"sd/{name}: rc={p.returncode} stdout={len(p.stdout)}B stderr={len(p.stderr)}B hits={None if hits is None else len(hits)}B")


def canon(k):
    r = rc(k)
    return k if k >= r else r


def make_sd_cases():
    if os.path.isdir(SD_CASES):
        shutil.rmtree(SD_CASES)
    rng = random.Random(0x5D5D)
    strain = rand_dna(rng, 1500)
    strain = strain[:700] + "N" + strain[701:]
    kmers = [strain[i:i + 31] for i in range(0, len(strain) - 30) if "N" not in strain[i:i + 31]]
    inf = [kmers[i] for i in range(0, len(kmers), 9)]                 # every 9th k-mer is "informative"

    def reads_from(n, seed, lens=(31, 60, 100, 151)):
        r = random.Random(seed)
        out = []
        for i in range(n):
            ln = r.choice(lens)
            if r.random() < 0.7:
                a = r.randrange(0, len(strain) - ln)
                s = strain[a:a + ln]
                if r.random() < 0.5:
                    s = rc(s.replace("N", "A"))
            else:
                s = rand_dna(r, ln)
            if i % 6 == 5:
                s = s[:r.choice((5, 20, 30))]                          # short read: carries the previous tallies
            if i % 10 == 3:
                s = s[:10] + "N" + s[11:]
            if i % 13 == 7:
                s = s.lower()
            out.append(s)
        return out

    def fasta(reads, tag):
        return "".join(f">{tag}{i}\n{r}\n" for i, r in enumerate(reads))

    def fastq(reads, tag):
        return "".join(f"@{tag}{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads))

    # ------------------------------------------------------------ batch list with everything
    d = os.path.join(SD_CASES, "batch")
    write(os.path.join(d, "strain.fa"), ">s\n" + wrap(strain, 70) + "\n")
    lines = ["# informative k-mers", "#second comment"] + inf[:40] + [rc(inf[40]), inf[41].lower(), inf[5], "ACGT",
             rand_dna(rng, 31), inf[42] + "A"] + inf[43:]
    write(os.path.join(d, "inf.txt.gz"), "\n".join(lines) + "\n", gz=True)
    r1, r2 = reads_from(60, 1), reads_from(60, 2)
    write(os.path.join(d, "pe_1.fa"), fasta(r1, "a"))
    write(os.path.join(d, "pe_2.fa"), fasta(r2, "b"))
    se = reads_from(50, 3)
    write(os.path.join(d, "se.fq.gz"), fastq(se, "s"), gz=True)
    il = reads_from(40, 4)
    write(os.path.join(d, "il.fa"), fasta(il, "i"))
    short2 = reads_from(30, 5)
    write(os.path.join(d, "pe_short2.fa"), fasta(short2[:20], "c"))           # PE2 runs out early (FASTA)
    write(os.path.join(d, "B.txt"), "PE\tpe_1.fa\tpe_2.fa\nSE\tse.fq.gz\n#comment line\nXX\tfoo\nPEI\til.fa\nse\tpe_1.fa\n"
                                    "PE\tpe_1.fa\nSE\nPE\tpe_2.fa\tpe_short2.fa\n")
    run_sd_case("batch", ["-r", "strain.fa", "-a", "inf.txt.gz", "-B", "B.txt", "-o", "hits.gz"],
                "SE/PE/PEI, comments and unknown types in -B, short reads inheriting tallies, PE2 ending early, "
                "odd lines in the informative list")

    # ------------------------------------------------------------ command-line forms
    for name, extra in (("cli_se", ["-b", "se.fq.gz", "-t", "SE"]), ("cli_pe", ["-b", "pe_1.fa", "-c", "pe_2.fa", "-t", "PE"]),
                        ("cli_pei", ["-b", "il.fa", "-t", "pei"]), ("cli_default", ["-b", "pe_1.fa"])):
        dd = os.path.join(SD_CASES, name)
        os.makedirs(dd, exist_ok=True)
        for f in ("strain.fa", "inf.txt.gz", "pe_1.fa", "pe_2.fa", "se.fq.gz", "il.fa"):
            shutil.copyfile(os.path.join(d, f), os.path.join(dd, f))
        run_sd_case(name, ["-r", "strain.fa", "-a", "inf.txt.gz"] + extra + ["-o", "out.gz"], "single metagenome on the command line")

    # ------------------------------------------------------------ background filter (-g)
    dd = os.path.join(SD_CASES, "background")
    os.makedirs(dd, exist_ok=True)
    for f in ("strain.fa", "pe_1.fa", "pe_2.fa", "se.fq.gz"):
        shutil.copyfile(os.path.join(d, f), os.path.join(dd, f))
    write(os.path.join(dd, "inf.txt"), "\n".join(inf) + "\n")
    bg = reads_from(80, 9, lens=(151,))
    write(os.path.join(dd, "bg1.fa"), fasta(bg[:40], "g"))
    write(os.path.join(dd, "bg2.fa"), fasta(bg[40:], "h"))
    write(os.path.join(dd, "bg.txt"), "bg1.fa\nbg2.fa\n")
    write(os.path.join(dd, "B.txt"), "PE\tpe_1.fa\tpe_2.fa\nSE\tse.fq.gz\n")
    run_sd_case("background", ["-r", "strain.fa", "-a", "inf.txt", "-g", "bg.txt", "-B", "B.txt", "-o", "hits.gz"],
                "-g demotes informative k-mers that are frequent in the background metagenomes")

    # ------------------------------------------------------------ argument errors
    dd = os.path.join(SD_CASES, "errors")
    os.makedirs(dd, exist_ok=True)
    for f in ("strain.fa", "inf.txt.gz", "pe_1.fa"):
        shutil.copyfile(os.path.join(d, f), os.path.join(dd, f))
    for name, argv in (("err_missing", ["-r", "strain.fa", "-a", "inf.txt.gz"]),
                       ("err_type", ["-r", "strain.fa", "-a", "inf.txt.gz", "-b", "pe_1.fa", "-t", "XX", "-o", "o.gz"]),
                       ("err_pe_one_file", ["-r", "strain.fa", "-a", "inf.txt.gz", "-b", "pe_1.fa", "-t", "PE", "-o", "o.gz"]),
                       ("err_b_and_B", ["-r", "strain.fa", "-a", "inf.txt.gz", "-b", "pe_1.fa", "-B", "x", "-o", "o.gz"]),
                       ("err_no_inf", ["-r", "strain.fa", "-a", "nope.gz", "-b", "pe_1.fa", "-o", "o.gz"]),
                       ("err_no_read1", ["-r", "strain.fa", "-a", "inf.txt.gz", "-b", "nope.fa", "-o", "o.gz"])):
        d3 = os.path.join(SD_CASES, name)
        shutil.copytree(dd, d3)
        run_sd_case(name, argv, "argument / file errors")
    shutil.rmtree(dd)


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    if os.path.isdir(CASES):
        shutil.rmtree(CASES)
    make_sd_cases()
    rng = random.Random(0x5EED31)

    # ---------------------------------------------------------------- case: mixed
    # one strain with upper/lower-case contigs, an N, a repeated segment; genomes list with a
    # duplicate line; metagenome reads in FASTA, FASTQ (quality starting with '@', multi-line),
    # gz, CRLF, interior N, 30/31-bp reads, reverse-complement reads, junk letters.
    d = os.path.join(CASES, "mixed")
    c1 = rand_dna(rng, 400)
    c2 = rand_dna(rng, 300)
    c2 = c2[:150] + "N" + c2[151:]
    c3 = c1[50:130] + rand_dna(rng, 60)          # shares 80 bp with contig 1 (multiplicity 2)
    strain = (">c1 first contig\n" + wrap(c1, 60) + "\n>c2\n" + wrap(c2.lower(), 70) +
              "\n>c3\tlast\n" + wrap(c3, 50) + "\n")
    write(os.path.join(d, "strain.fna.gz"), strain, gz=True)
    g1 = rand_dna(rng, 200) + c1[100:300] + rand_dna(rng, 100)
    g2 = rc(c2[160:290]) + rand_dna(rng, 120)
    write(os.path.join(d, "g1.fa"), ">g1\n" + wrap(g1, 80) + "\n")
    write(os.path.join(d, "g2.fa.gz"), ">g2 x\n" + wrap(g2, 61) + "\n\n>g2b\n" + c3[:90] + "\n", gz=True)
    write(os.path.join(d, "A.txt"), "g1.fa\ng2.fa.gz\ng1.fa\n")     # duplicate line counted twice
    reads = []
    for i in range(40):
        src = rng.choice([c1, c2, c3])
        a = rng.randrange(0, len(src) - 100)
        r = src[a:a + rng.choice([31, 45, 75, 100])]
        if i % 3 == 0:
            r = rc(r.replace("N", "A"))
        if i % 7 == 0:
            r = r[:20] + "N" + r[21:]
        if i % 11 == 0:
            r = r.lower()
        reads.append(r)
    reads.append(c1[10:40])                       # 30 bp: skipped (shorter than k)
    reads.append(c1[10:41])                       # 31 bp: exactly one window
    reads.append(c1[200:240] + "R" + c1[241:300]) # IUPAC letter in a read
    reads.append(c1[300:330] + "-" + c1[331:380])
    reads.append(c1[100:160].replace("T", "U"))   # U in a read
    reads.append(rc(c1[100:160]).replace("T", "U"))
    fa = "".join(f">r{i} len={len(r)}\n{wrap(r, 50)}\n" for i, r in enumerate(reads))
    write(os.path.join(d, "m1.fasta"), fa)
    fq = []
    for i, r in enumerate(reads[:25]):
        q = "@" + "I" * (len(r) - 1)              # quality line starting with '@'
        fq.append(f"@q{i}/1\n{r}\n+\n{q}\n")
    r = reads[5]
    fq.append(f"@multi\n{r[:20]}\n{r[20:]}\n+multi\n{'I' * 20}\n{'I' * (len(r) - 20)}\n")
    write(os.path.join(d, "m2.fq.gz"), "".join(fq), gz=True)
    crlf = "".join(f">w{i}\r\n{wrap(r, 40).replace(chr(10), chr(13) + chr(10))}\r\n" for i, r in enumerate(reads[:12]))
    write(os.path.join(d, "m3_crlf.fa"), crlf)
    write(os.path.join(d, "B.txt"), "m1.fasta\nm2.fq.gz\nm3_crlf.fa")  # no trailing newline
    run_case("mixed", ["-r", "strain.fna.gz", "-A", "A.txt", "-B", "B.txt", "-p", "progress.txt"],
             "4-field rows, constant 5-name header, duplicate list line, FASTQ/CRLF/gz/N/U/IUPAC-in-read")

    # ---------------------------------------------------------------- case: drug (-C with skip)
    d2 = os.path.join(CASES, "drug")
    os.makedirs(d2, exist_ok=True)
    for f in ("strain.fna.gz", "g1.fa", "g2.fa.gz", "m1.fasta", "A.txt"):
        shutil.copyfile(os.path.join(d, f), os.path.join(d2, f))
    write(os.path.join(d2, "B.txt"), "m1.fasta\n")
    write(os.path.join(d2, "C.txt"), "g2.fa.gz\nstrain.fna.gz\n./strain.fna.gz\n")
    run_case("drug", ["-r", "strain.fna.gz", "-A", "A.txt", "-B", "B.txt", "-C", "C.txt", "-p", "prog.txt"],
             "5-field rows; -C line equal to -r is skipped (stderr), './strain.fna.gz' is NOT skipped")

    # ---------------------------------------------------------------- case: iupac_strain
    d3 = os.path.join(CASES, "iupac_strain")
    s = rand_dna(rng, 120)
    s_i = s[:40] + "R" + s[41:80] + "K" + s[81:]
    write(os.path.join(d3, "strain.fa"), ">s\n" + s_i + "\n")
    rd = [s_i[20:70], rc(s[20:70]), s[20:70], s_i[60:110], s[60:80] + "M" + s[81:110],
          rc(s[:40]) , s_i[30:62].replace("R", "Y")]
    write(os.path.join(d3, "m.fa"), "".join(f">r{i}\n{r}\n" for i, r in enumerate(rd)))
    write(os.path.join(d3, "A.txt"), "m.fa\n")
    write(os.path.join(d3, "B.txt"), "m.fa\n")
    run_case("iupac_strain", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt"],
             "non-N IUPAC letters in the strain become keys; K complements to '.'")

    # ---------------------------------------------------------------- case: truncated_fastq
    d4 = os.path.join(CASES, "truncated_fastq")
    s = rand_dna(rng, 200)
    write(os.path.join(d4, "strain.fa"), ">s\n" + s + "\n")
    fq = (f"@a\n{s[0:50]}\n+\n{'I' * 50}\n@b\n{s[50:100]}\n+\n{'I' * 30}\n@c\n{s[100:150]}\n+\n{'I' * 50}\n")
    write(os.path.join(d4, "m.fq"), fq)
    write(os.path.join(d4, "n.fa"), f">x\n{s[150:200]}\n")
    write(os.path.join(d4, "A.txt"), "n.fa\n")
    write(os.path.join(d4, "B.txt"), "m.fq\nn.fa\n")
    run_case("truncated_fastq", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt"],
             "record b has a short quality: parser swallows the next header looking for quality, "
             "length mismatch ends that file silently; later files still scanned")

    # ---------------------------------------------------------------- error paths
    d5 = os.path.join(CASES, "missing_in_list")
    write(os.path.join(d5, "strain.fa"), ">s\n" + rand_dna(rng, 100) + "\n")
    write(os.path.join(d5, "A.txt"), "nope.fa\n")
    write(os.path.join(d5, "B.txt"), "nope.fa\n")
    run_case("missing_in_list", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt"], "exit 1 + message")

    d6 = os.path.join(CASES, "missing_flag")
    write(os.path.join(d6, "strain.fa"), ">s\n" + rand_dna(rng, 100) + "\n")
    run_case("missing_flag", ["-r", "strain.fa", "-A", "A.txt"], "usage on stderr, exit 1")

    d7 = os.path.join(CASES, "short_contig")
    write(os.path.join(d7, "strain.fa"), ">s\n" + rand_dna(rng, 100) + "\n>tiny\nACGTACGTAC\n")
    write(os.path.join(d7, "A.txt"), "strain.fa\n")
    write(os.path.join(d7, "B.txt"), "strain.fa\n")
    run_case("short_contig", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt"],
             "reference crashes (SIGSEGV) on a strain record shorter than k-1; the product skips it")

    d8 = os.path.join(CASES, "contig30")
    s = rand_dna(rng, 100)
    write(os.path.join(d8, "strain.fa"), ">s\n" + s + "\n>thirty\n" + rand_dna(rng, 30) + "\n")
    write(os.path.join(d8, "A.txt"), "strain.fa\n")
    write(os.path.join(d8, "B.txt"), "strain.fa\nstrain.fa\n")
    run_case("contig30", ["-r", "strain.fa", "-A", "A.txt", "-B", "B.txt", "-h"],
             "a 30-bp strain record yields zero windows and no crash; -h prints usage and carries on")

    # ---------------------------------------------------------------- bundled example (cfg 1)
    os.makedirs(BUNDLED, exist_ok=True)
    for sub in ("strains", "metagenomes"):
        os.makedirs(os.path.join(BUNDLED, sub), exist_ok=True)
        for f in sorted(os.listdir(os.path.join(REF_TEST, sub))):
            dst = os.path.join(BUNDLED, sub, f)
            shutil.copyfile(os.path.join(REF_TEST, sub, f), dst)
            os.chmod(dst, 0o644)
    for f in ("genomes_to_scrub.txt", "metagenomes_to_scrub.txt", "target_metagenomes.txt"):
        shutil.copyfile(os.path.join(REF_TEST, f), os.path.join(BUNDLED, f))
        os.chmod(os.path.join(BUNDLED, f), 0o644)
    strain = "strains/Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208.fna.gz"
    argv = ["-r", strain, "-A", "genomes_to_scrub.txt", "-B", "metagenomes_to_scrub.txt"]
    p = subprocess.run([REF_BIN] + argv, cwd=BUNDLED, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = p.stdout
    lines = out.split(b"\n")
    col_sums = [0, 0, 0]
    nz = [0, 0, 0]
    for ln in lines[1:]:
        if not ln:
            continue
        f = ln.split(b"\t")
        for j in range(3):
            v = int(f[1 + j])
            col_sums[j] += v
            nz[j] += v > (1 if j == 0 else 0)
    facts = {"argv": argv, "returncode": p.returncode, "stdout_bytes": len(out),
             "stdout_lines": out.count(b"\n"), "stdout_md5": hashlib.md5(out).hexdigest(),
             "first_rows": [l.decode() for l in lines[:6]],
             "last_rows": [l.decode() for l in lines[-4:-1]],
             "column_sums": col_sums,
             "rows_with_ref_gt1_pan_gt0_meta_gt0": nz}
    with open(os.path.join(BUNDLED, "step1_facts.json"), "w") as f:
        json.dump(facts, f, indent=1)
    # step 3 (strain_detect) needs step 2's k-mer list: strains/B8.scrubbed_kmers.gz is the output of the
    # reference's scripts/kmer_scrub_filter.py -m 0.01 on the step-1 TSV (md5 fe981fa5..., generated once
    # in the build container and kept as a data fixture)
    scrub = "strains/B8.scrubbed_kmers.gz"
    if os.path.exists(os.path.join(BUNDLED, scrub)):
        argv3 = ["-r", strain, "-a", scrub, "-B", "target_metagenomes.txt", "-o", "step3_hits.gz"]
        p3 = subprocess.run([REF_SD] + argv3, cwd=BUNDLED, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        with gzip.open(os.path.join(BUNDLED, "step3_hits.gz"), "rb") as f:
            hits = f.read()
        os.remove(os.path.join(BUNDLED, "step3_hits.gz"))
        with open(os.path.join(BUNDLED, "step3_expected.hits"), "wb") as f:
            f.write(hits)
        with open(os.path.join(BUNDLED, "step3_facts.json"), "w") as f:
            json.dump({"argv": argv3, "returncode": p3.returncode, "stdout": p3.stdout.decode(), "stderr": p3.stderr.decode(),
                       "hits_md5": hashlib.md5(hits).hexdigest(), "hits_lines": hits.count(b"\n")}, f, indent=1)
        print("bundled step 3:", hashlib.md5(hits).hexdigest(), hits.count(b"\n"))
    print("bundled:", facts["stdout_md5"], facts["stdout_lines"], col_sums, nz)


if __name__ == "__main__":
    main()
