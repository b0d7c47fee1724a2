#!/usr/bin/env python3
"""Generate the golden fixtures for the two consumers either side of the scan:
tests/golden/filter_cases/ (scripts/kmer_scrub_filter.py) and tests/golden/cov_cases/
(scripts/coverage_depth.py).

Run HERE (the build container), never on the GPU box:

    python tests/golden/make_golden_filter.py

Every expected output is produced by RUNNING the reference's own Python scripts from
/root/reference/scripts (as subprocesses; nothing of them is copied).  The fixtures are DATA: small
synthetic count tables / hit files written by this script with a fixed seed, and the bytes the
reference scripts printed for them.
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
FILTER_REF = "/root/reference/scripts/kmer_scrub_filter.py"
COV_REF = "/root/reference/scripts/coverage_depth.py"
FCASES = os.path.join(HERE, "filter_cases")
CCASES = os.path.join(HERE, "cov_cases")
HEADER = "#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n"


def rand_kmer(rng, k=31):
    return "".join(rng.choice("ACGT") for _ in range(k))


def write_gz(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(text.encode("latin-1"))


def write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def table(rows, header=True):
    out = [HEADER] if header else []
    for r in rows:
        out.append("\t".join(str(x) for x in r) + "\n")
    return "".join(out)


def run(ref, root, name, argv, note, stderr_exact=False):
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    p = subprocess.run([sys.executable, ref] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(os.path.join(d, "expected.stdout"), "wb") as f:
        f.write(p.stdout)
    err = p.stderr
    if not stderr_exact:                       # tracebacks carry paths: keep only the last line
        lines = err.decode("latin-1").splitlines()
        err = (lines[-1] + "\n").encode("latin-1") if lines else b""
    with open(os.path.join(d, "expected.stderr"), "wb") as f:
        f.write(err)
    with open(os.path.join(d, "case.json"), "w") as f:
        json.dump({"argv": argv, "returncode": p.returncode, "note": note, "stderr_exact": stderr_exact}, f, indent=1)
    print(f"{os.path.basename(root)}/{name}: rc={p.returncode} stdout={len(p.stdout)}B stderr={len(p.stderr)}B")


def skewed(rng, zero_frac, hi):
    if rng.random() < zero_frac:
        return 0
    r = rng.random()
    if r < 0.6:
        return rng.randrange(1, 4)
    if r < 0.9:
        return rng.randrange(1, 40)
    return rng.randrange(1, hi)


def make_filter_cases():
    if os.path.isdir(FCASES):
        shutil.rmtree(FCASES)
    rng = random.Random(0xF117E2)

    # ---- joint scrub, 4 columns, heavy ties and zeros, a range of -m
    keys = [rand_kmer(rng) for _ in range(3000)]
    rows = [(k, rng.randrange(1, 3), skewed(rng, 0.5, 5000), skewed(rng, 0.3, 90000)) for k in keys]
    for name, argv, note in (
        ("joint_default", ["-s", "t.gz"], "default -m 0.04"),
        ("joint_m_half", ["-s", "t.gz", "-m", "0.5"], "-m 0.5"),
        ("joint_m0", ["-s", "t.gz", "-m", "0"], "-m 0 leaves exactly one k-mer"),
        ("joint_m1", ["-s", "t.gz", "-m", "1.0"], "-m 1 scrubs nothing"),
        ("joint_m_small", ["-s", "t.gz", "-m", "0.0123"], "-m 0.0123"),
        ("joint_long_opts", ["--scrub_count_file", "t.gz", "--min_fraction=0.25"], "long options"),
        ("independent", ["-s", "t.gz", "-m", "0.3", "-i"], "independent scrub; stderr lists every threshold"),
        ("independent_default", ["-s", "t.gz", "--independent"], "independent, default -m"),
        ("independent_high", ["-s", "t.gz", "-m", "0.9", "-i"], "independent, many thresholds"),
    ):
        write_gz(os.path.join(FCASES, name, "t.gz"), table(rows))
        run(FILTER_REF, FCASES, name, argv, note, stderr_exact=name.startswith("independent"))

    # ---- the cut falls inside a long run of equal scores (stable order decides)
    keys = [rand_kmer(rng) for _ in range(1000)]
    rows = [(k, 1, 7 if i % 3 else 0, 7 if i % 2 else 3) for i, k in enumerate(keys)]
    for m in ("0.2", "0.5", "0.8"):
        name = "ties_m" + m.replace(".", "_")
        write_gz(os.path.join(FCASES, name, "t.gz"), table(rows))
        run(FILTER_REF, FCASES, name, ["-s", "t.gz", "-m", m], "cut inside a run of equal scores")

    # ---- all counts zero: no normalisation at all, everything ties at 0
    rows = [(rand_kmer(rng), 1, 0, 0) for _ in range(200)]
    write_gz(os.path.join(FCASES, "all_zero", "t.gz"), table(rows))
    run(FILTER_REF, FCASES, "all_zero", ["-s", "t.gz", "-m", "0.5"], "both sums zero")

    # ---- counters >= 2^31 were printed negative by %d: never > 0, never summed
    rows = [(rand_kmer(rng), 1, rng.choice([0, 5, -2147483648, -1, 12]), rng.choice([0, 3, -7, 900, 2147483647]))
            for _ in range(400)]
    write_gz(os.path.join(FCASES, "negatives", "t.gz"), table(rows))
    run(FILTER_REF, FCASES, "negatives", ["-s", "t.gz", "-m", "0.4"], "negative fields, INT_MAX")
    write_gz(os.path.join(FCASES, "negatives_i", "t.gz"), table(rows))
    run(FILTER_REF, FCASES, "negatives_i", ["-s", "t.gz", "-m", "0.1", "-i"], "negative fields, independent", True)

    # ---- 5 columns: drug scrub first
    keys = [rand_kmer(rng) for _ in range(1500)]
    rows = [(k, 1, skewed(rng, 0.5, 300), skewed(rng, 0.4, 3000), 1 if rng.random() < 0.15 else 0) for k in keys]
    write_gz(os.path.join(FCASES, "drug_ok", "t.gz"), table(rows))
    run(FILTER_REF, FCASES, "drug_ok", ["-s", "t.gz", "-m", "0.1"], "5 fields, 15 % cross-drug")
    write_gz(os.path.join(FCASES, "drug_ok_i", "t.gz"), table(rows))
    run(FILTER_REF, FCASES, "drug_ok_i", ["-s", "t.gz", "-m", "0.1", "-i"], "5 fields, independent", True)
    rows_bad = [(k, 1, a, b, 1 if rng.random() < 0.9 else 0) for (k, _, a, b, _) in rows]
    write_gz(os.path.join(FCASES, "drug_fail", "t.gz"), table(rows_bad))
    run(FILTER_REF, FCASES, "drug_fail", ["-s", "t.gz", "-m", "0.1"], "too few remain after the drug scrub: exception, rc 1")
    # a file where only some lines carry the 5th field
    rows_mixed = [r if i % 4 else r[:4] for i, r in enumerate(rows)]
    write_gz(os.path.join(FCASES, "drug_mixed_cols", "t.gz"), table(rows_mixed))
    run(FILTER_REF, FCASES, "drug_mixed_cols", ["-s", "t.gz", "-m", "0.2"], "4- and 5-field lines in one file")

    # ---- comments inside, duplicate keys (line count vs distinct count), no header
    keys = [rand_kmer(rng) for _ in range(300)]
    rows = [(k, 1, skewed(rng, 0.4, 50), skewed(rng, 0.4, 500)) for k in keys]
    rows += [(keys[i], 2, 4, 9) for i in (3, 3, 17, 250)]
    text = table(rows[:100], header=False) + "# a comment line\n" + table(rows[100:], header=False) + "#tail\n"
    write_gz(os.path.join(FCASES, "comments_dups", "t.gz"), text)
    run(FILTER_REF, FCASES, "comments_dups", ["-s", "t.gz", "-m", "0.3"], "comments, duplicate keys, no header")
    write_gz(os.path.join(FCASES, "comments_dups_i", "t.gz"), text)
    run(FILTER_REF, FCASES, "comments_dups_i", ["-s", "t.gz", "-m", "0.3", "-i"], "same, independent", True)
    dup5 = "".join(ln + ("\t1\n" if i % 9 == 0 else "\t0\n") for i, ln in enumerate(table(rows, header=False).splitlines()))
    write_gz(os.path.join(FCASES, "dups_drug", "t.gz"), HEADER + dup5)
    run(FILTER_REF, FCASES, "dups_drug", ["-s", "t.gz", "-m", "0.2"], "duplicates + drug: drug_scrubbed counts lines")

    # ---- header only / no data
    write_gz(os.path.join(FCASES, "empty", "t.gz"), HEADER)
    run(FILTER_REF, FCASES, "empty", ["-s", "t.gz"], "no rows")
    write_gz(os.path.join(FCASES, "empty_i", "t.gz"), HEADER)
    run(FILTER_REF, FCASES, "empty_i", ["-s", "t.gz", "-i"], "no rows, independent: division by zero, rc 1")
    one = [(rand_kmer(rng), 1, 3, 4)]
    write_gz(os.path.join(FCASES, "one_row", "t.gz"), table(one))
    run(FILTER_REF, FCASES, "one_row", ["-s", "t.gz", "-m", "0.5"], "single row")

    # ---- list mode: counts add up over files; only files after the second are compared
    keys = [rand_kmer(rng) for _ in range(500)]
    def tab(seed):
        r2 = random.Random(seed)
        return table([(k, 1, skewed(r2, 0.5, 100), skewed(r2, 0.3, 1000)) for k in keys])
    for name, files, lst, note in (
        ("list_one", 1, "a.gz\n", "list with one file"),
        ("list_two", 2, "a.gz\nb.gz\n", "two files: counts add"),
        ("list_three", 3, "a.gz\nb.gz\nc.gz", "three files, no trailing newline"),
    ):
        for i in range(files):
            write_gz(os.path.join(FCASES, name, "abc"[i] + ".gz"), tab(100 + i))
        write(os.path.join(FCASES, name, "list.txt"), lst)
        run(FILTER_REF, FCASES, name, ["-l", "list.txt", "-m", "0.2"], note)
    name = "list_three_mismatch"
    write_gz(os.path.join(FCASES, name, "a.gz"), tab(1))
    write_gz(os.path.join(FCASES, name, "b.gz"), tab(2))
    write_gz(os.path.join(FCASES, name, "c.gz"), table([(k, 1, 1, 1) for k in keys[:-1]]))
    write(os.path.join(FCASES, name, "list.txt"), "a.gz\nb.gz\nc.gz\n")
    run(FILTER_REF, FCASES, name, ["-l", "list.txt"], "third file has a different key set: exit message, rc 1", True)
    name = "list_second_differs"
    write_gz(os.path.join(FCASES, name, "a.gz"), tab(1))
    write_gz(os.path.join(FCASES, name, "b.gz"), table([(k, 1, 2, 5) for k in keys[:300]] + [(rand_kmer(rng), 1, 9, 9)]))
    write(os.path.join(FCASES, name, "list.txt"), "a.gz\nb.gz\n")
    run(FILTER_REF, FCASES, name, ["-l", "list.txt", "-m", "0.3"],
        "the second file is never compared with the first; its rows are the strain, the first file's counts still add")
    write(os.path.join(FCASES, "list_i", "list.txt"), "a.gz\nb.gz\n")
    write_gz(os.path.join(FCASES, "list_i", "a.gz"), tab(5))
    write_gz(os.path.join(FCASES, "list_i", "b.gz"), tab(6))
    run(FILTER_REF, FCASES, "list_i", ["-l", "list.txt", "-m", "0.3", "-i"], "list, independent", True)

    # ---- argument handling
    os.makedirs(os.path.join(FCASES, "no_input"), exist_ok=True)
    run(FILTER_REF, FCASES, "no_input", [], "neither -s nor -l: message, then an empty result, rc 0", True)
    write_gz(os.path.join(FCASES, "bad_m", "t.gz"), table(one))
    run(FILTER_REF, FCASES, "bad_m", ["-s", "t.gz", "-m", "1.5"], "out-of-range -m: the script dies building its message, rc 1")
    write_gz(os.path.join(FCASES, "both_inputs", "t.gz"), table(one))
    write(os.path.join(FCASES, "both_inputs", "list.txt"), "t.gz\n")
    run(FILTER_REF, FCASES, "both_inputs", ["-s", "t.gz", "-l", "list.txt", "-m", "0.5"], "both -s and -l: message, -s wins", True)
    os.makedirs(os.path.join(FCASES, "missing_file"), exist_ok=True)
    run(FILTER_REF, FCASES, "missing_file", ["-s", "nope.gz"], "unreadable input: rc 1")
    write_gz(os.path.join(FCASES, "malformed", "t.gz"), table(one) + "ACGT\n")
    run(FILTER_REF, FCASES, "malformed", ["-s", "t.gz"], "line with too few fields: rc 1, nothing on stdout")

    # ---- real tables from step 1 of the scan goldens
    for src in ("mixed", "drug"):
        with open(os.path.join(HERE, "cases", src, "expected.stdout"), "rb") as f:
            data = f.read().decode("latin-1")
        name = "from_scan_" + src
        write_gz(os.path.join(FCASES, name, "t.gz"), data)
        run(FILTER_REF, FCASES, name, ["-s", "t.gz", "-m", "0.3" if src == "mixed" else "0.0"],
            "table printed by kmer_scrub_count for tests/golden/cases/" + src)


def hits_file(rng, samples, kmers, with_stats=True, dup_rate=0.3):
    out = []
    for s in samples:
        n = rng.randrange(0, 60)
        for _ in range(n):
            a, b = rng.randrange(0, 30), rng.randrange(0, 6)
            c, d = rng.choice([0, 0, rng.randrange(0, 30)]), rng.randrange(0, 4)
            k = rng.choice(kmers[:10]) if rng.random() < dup_rate else rng.choice(kmers)
            out.append(f"{s}\t{a}\t{b}\t{c}\t{d}\t{k}\n")
        if with_stats:
            out.append(f"#{s}\ttotal_kmer_evaluated\t{rng.randrange(0, 10**9)}\n")
            out.append(f"#{s}\ttotal_reads_evaluated\t{rng.randrange(0, 10**7)}\n")
            out.append(f"#{s}\ttotal_genome_kmers\t{len(kmers) * 50}\n")
            out.append(f"#{s}\ttotal_genome_informative_kmers\t{len(kmers)}\n")
    return "".join(out)


def make_cov_cases():
    if os.path.isdir(CCASES):
        shutil.rmtree(CCASES)
    rng = random.Random(0xC0FE)
    kmers = [rand_kmer(rng) for _ in range(120)]
    samples = ["metagenomes/s1_PE1.fastq.gz", "/abs/path/s2.fasta.gz", "s3.fq", "dir/s1_PE1.fastq.gz"]
    text = hits_file(rng, samples, kmers)
    for name, argv, note in (
        ("basic", ["-k", "Genus_species_strain1.kmer_hits.gz"], "default -m 1"),
        ("min3", ["-k", "Genus_species_strain1.kmer_hits.gz", "-m", "3"], "-m 3"),
        ("min0", ["-k", "Genus_species_strain1.kmer_hits.gz", "--min_kmer_hits", "0"], "-m 0"),
        ("background", ["-k", "Genus_species_strain1.kmer_hits.gz", "-b", "bg.txt"], "background list"),
    ):
        write_gz(os.path.join(CCASES, name, "Genus_species_strain1.kmer_hits.gz"), text)
        write(os.path.join(CCASES, name, "bg.txt"), "s3.fq\ns2.fasta.gz\nnot_there\n")
        run(COV_REF, CCASES, name, argv, note, True)
    # one-word strain name, sample with zero hits but stats, zero evaluated k-mers
    t2 = ("#m0.fa\ttotal_kmer_evaluated\t0\n#m0.fa\ttotal_reads_evaluated\t0\n#m0.fa\ttotal_genome_kmers\t9\n"
          "#m0.fa\ttotal_genome_informative_kmers\t3\n" + hits_file(rng, ["m1.fa"], kmers))
    write_gz(os.path.join(CCASES, "oneword", "strainX.kmer_hits.gz"), t2)
    run(COV_REF, CCASES, "oneword", ["-k", "strainX.kmer_hits.gz"], "sample without hit lines; zero evaluated k-mers", True)
    # hit lines but the stats block is missing: informative count -1 in the ratio
    write_gz(os.path.join(CCASES, "no_stats", "a_b.kmer_hits.gz"), hits_file(rng, ["q.fa"], kmers, with_stats=False))
    run(COV_REF, CCASES, "no_stats", ["-k", "a_b.kmer_hits.gz"], "no trailer lines: -1 placeholders", True)
    # informative total 0 -> division by zero
    t3 = "x.fa\t5\t1\t0\t0\t" + kmers[0] + "\n#x.fa\ttotal_kmer_evaluated\t10\n#x.fa\ttotal_reads_evaluated\t1\n" \
         "#x.fa\ttotal_genome_kmers\t0\n#x.fa\ttotal_genome_informative_kmers\t0\n"
    write_gz(os.path.join(CCASES, "zero_informative", "a.kmer_hits.gz"), t3)
    run(COV_REF, CCASES, "zero_informative", ["-k", "a.kmer_hits.gz"], "float division by zero: rc 1 after the header")
    # k-mer text outside strain_detect's alphabet / of mixed length; two different (sample, k-mer) pairs
    # that join to the same string ("s1"+"ACGTACGT" == "s1A"+"CGTACGT"): the script counts the second as seen
    t4 = ("s1\t5\t1\t0\t0\tACGTACGT\n" "d/s1A\t5\t1\t0\t0\tCGTACGT\n" "s1A\t9\t1\t0\t0\tCGTACGT\n"
          "s1\t5\t1\t3\t0\tacgtnnrya\n" "s1\t5\t1\t3\t0\tacgtnnrya\n" "s2\t2\t1\t0\t0\t" + kmers[3] + "\n"
          "s2\t1\t1\t0\t0\t" + kmers[4] + "\n" "s2\t7\t1\t0\t0\t" + kmers[3] + "\textra\tfields\n")
    for smp in ("s1", "s1A", "s2"):
        t4 += f"#{smp}\ttotal_kmer_evaluated\t1000\n#{smp}\ttotal_reads_evaluated\t10\n#{smp}\ttotal_genome_kmers\t99\n#{smp}\ttotal_genome_informative_kmers\t7\n"
    write_gz(os.path.join(CCASES, "general_text", "x_y_z.kmer_hits.gz"), t4)
    run(COV_REF, CCASES, "general_text", ["-k", "x_y_z.kmer_hits.gz"], "ragged / non-ACGT k-mer text and a joined-string collision", True)
    # trailer-only samples are listed in the order of their total_kmer_evaluated lines
    t5 = ("#x.fa\ttotal_reads_evaluated\t5\n#y.fa\ttotal_kmer_evaluated\t50\n#x.fa\ttotal_kmer_evaluated\t40\n"
          "#x.fa\ttotal_genome_informative_kmers\t4\n#y.fa\ttotal_genome_informative_kmers\t4\n"
          "#x.fa\ttotal_genome_kmers\t8\n#y.fa\ttotal_genome_kmers\t8\n#z.fa\ttotal_reads_evaluated\t1\t \n")
    write_gz(os.path.join(CCASES, "trailer_order", "g.kmer_hits.gz"), t5)
    run(COV_REF, CCASES, "trailer_order", ["-k", "g.kmer_hits.gz"], "order of trailer-only samples; trailing blanks", True)
    write_gz(os.path.join(CCASES, "short_line", "g.kmer_hits.gz"), "s\t1\t1\t1\t1\n")
    run(COV_REF, CCASES, "short_line", ["-k", "g.kmer_hits.gz"], "five fields: IndexError, rc 1")
    write_gz(os.path.join(CCASES, "empty", "a_b_c.kmer_hits.gz"), "")
    run(COV_REF, CCASES, "empty", ["-k", "a_b_c.kmer_hits.gz"], "empty hits file: header only", True)
    # the real step-3 golden of the bundled example
    src = os.path.join(HERE, "bundled", "step3_expected.hits")
    if os.path.exists(src):
        with open(src, "rb") as f:
            data = f.read().decode("latin-1")
        nm = "Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208.kmer_hits.gz"
        write_gz(os.path.join(CCASES, "bundled_step4", nm), data)
        run(COV_REF, CCASES, "bundled_step4", ["-k", nm], "test/example.sh step 4 on the step-3 golden", True)
        os.remove(os.path.join(CCASES, "bundled_step4", nm))      # the input is tests/golden/bundled/step3_expected.hits


if __name__ == "__main__":
    for ref in (FILTER_REF, COV_REF):
        if not os.path.exists(ref):
            sys.exit("reference scripts not found (run this in the build container)")
    make_filter_cases()
    make_cov_cases()
