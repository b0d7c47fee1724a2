#!/usr/bin/env python3
"""Wall clock of the reference's four-step workflow (test/example.sh) on the bundled example data kept under
tests/golden/bundled: the GPU programs of this repo, step by step and with steps 1+2 fused, next to
the CPU restatements (oracle/) of the same steps where they have been built.  Output checks are md5s of
the decompressed results against the recorded reference facts.  Prints one JSON object.

The reference's own programs cannot travel to the GPU box; their times, measured in the build container,
are in BASELINE.md."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = os.path.join(REPO, "tests", "golden", "bundled")
BIN = os.path.join(REPO, "strainer2_amd", "bin")
ORA = os.path.join(REPO, "oracle")
STRAIN = "strains/Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208.fna.gz"
NM = "Bacteroides_ovatus_1001283st1_B8_1001283B150210_160208"


def timed(cmd, stdout=None, **kw):
    t = time.time()
    p = subprocess.run(cmd, cwd=B, stdout=stdout, stderr=subprocess.PIPE, **kw)
    assert p.returncode == 0, (cmd, p.stderr[-500:])
    return time.time() - t


def md5_gz(path):
    h = hashlib.md5()
    with gzip.open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def pipe_gz(cmd, dst, level="-1"):
    t = time.time()
    p1 = subprocess.Popen(cmd, cwd=B, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(dst, "wb") as f:
        p2 = subprocess.run(["gzip", level], stdin=p1.stdout, stdout=f)
    p1.stdout.close()
    assert p1.wait() == 0 and p2.returncode == 0, p1.stderr.read()[-500:]
    return time.time() - t


def run_set(tag, step1, step2, step3, step4, work, out, fused=None):
    os.makedirs(os.path.join(work, tag))
    counts, scrub, hits, cov = (os.path.join(work, tag, x) for x in ("counts.gz", "scrubbed.gz", NM + ".kmer_hits.gz", "cov.tsv"))
    r = {}
    r["step1_count_s"] = pipe_gz([step1, "-r", STRAIN, "-A", "genomes_to_scrub.txt", "-B", "metagenomes_to_scrub.txt"], counts)
    r["step2_filter_s"] = pipe_gz([step2, "-s", counts, "-m", "0.01"], scrub)
    r["step3_detect_s"] = timed([step3, "-r", STRAIN, "-a", scrub, "-B", "target_metagenomes.txt", "-o", hits])
    with open(cov, "wb") as f:
        r["step4_coverage_s"] = timed([step4, "-k", hits], stdout=f)
    r["total_s"] = sum(r.values())
    r["md5_counts"] = md5_gz(counts)
    r["md5_scrubbed"] = md5_gz(scrub)
    r["md5_hits"] = md5_gz(hits)
    r["md5_coverage"] = hashlib.md5(open(cov, "rb").read()).hexdigest()
    if fused:
        f2 = os.path.join(work, tag, "fused.gz")
        r["steps1+2_fused_s"] = pipe_gz([step1, "-r", STRAIN, "-A", "genomes_to_scrub.txt", "-B", "metagenomes_to_scrub.txt", "--scrub", "0.01"], f2)
        r["md5_fused"] = md5_gz(f2)
        cov2 = os.path.join(work, tag, NM + ".coverage_depth")
        hits2 = os.path.join(work, tag, "f", NM + ".kmer_hits.gz")
        os.makedirs(os.path.dirname(hits2))
        r["steps3+4_fused_s"] = timed([step3, "-r", STRAIN, "-a", f2, "-B", "target_metagenomes.txt", "-o", hits2, "--coverage-depth=" + cov2])
        r["md5_fused_coverage"] = hashlib.md5(open(cov2, "rb").read()).hexdigest()
        r["total_fused_s"] = r["steps1+2_fused_s"] + r["steps3+4_fused_s"]
    out[tag] = r


def main():
    out = {"workload": "test/example.sh on the bundled data (6.7 Mbp strain; 3 genomes, 2+2 metagenomes), gzip -1 between steps"}
    with tempfile.TemporaryDirectory() as work:
        run_set("gpu", *(os.path.join(BIN, x) for x in ("kmer_scrub_count", "kmer_scrub_filter", "strain_detect", "coverage_depth")),
                work, out, fused=True)
        run_set("gpu_second_run", *(os.path.join(BIN, x) for x in ("kmer_scrub_count", "kmer_scrub_filter", "strain_detect", "coverage_depth")),
                work, out, fused=True)
        ora = [os.path.join(ORA, x) for x in ("kso_oracle", "ksf_oracle", "ksd_oracle", "kcd_oracle")]
        if all(os.path.exists(x) for x in ora) and "--no-cpu" not in sys.argv:
            run_set("cpu_restatement_1core", *ora, work, out)
    want = {"md5_counts": "75989a9bc31ef0b6f53a5112a60920bd", "md5_scrubbed": "fe981fa571be70e602875ac3463ecdac",
            "md5_hits": "e1799e705d4f693240573da32540efcc"}
    want_cov = hashlib.md5(open(os.path.join(REPO, "tests", "golden", "cov_cases", "bundled_step4", "expected.stdout"), "rb").read()).hexdigest()
    for tag, r in out.items():
        if isinstance(r, dict):
            r["outputs_match_reference"] = all(r[k] == v for k, v in want.items()) and r["md5_coverage"] == want_cov and \
                r.get("md5_fused", want["md5_scrubbed"]) == want["md5_scrubbed"] and r.get("md5_fused_coverage", want_cov) == want_cov
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
