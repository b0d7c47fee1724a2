#!/usr/bin/env python3
"""tests/golden/make_cfg5_share_facts.py -- pin one GPU's share of BASELINE configs[4] (32 strains resident, SE metagenome) to the
unmodified reference on a 1 Gbase prefix of the metagenome.

The job is strainer2_amd/cfg5.py.  The UNMODIFIED reference program (oracle/_ref/strain_detect, built by oracle/Makefile from
/root/reference/src) runs here, one process per pinned strain (cfg5.PINNED_STRAINS), on the first 6,666,667 reads (1.0 Gbase) of the
metagenome: `strain_detect -r s<k>.fa -a s<k>.inf -b prefix.fa -t SE -o <out>` (src/strain_detect.c:61-158,387-663); recorded per
strain: md5 / lines / bytes of the DECOMPRESSED -o file (hit lines and the four '#' trailer lines), stdout and stderr.

tools/sd_cfg5_share.py writes the same inputs on the GPU box, runs bin/strain_detect -S with all 32 strains on the prefix and
compares these two strains' files; then it runs the full 100 Gbase pass.  Only data is committed
(tests/golden/cfg5_share_facts.json).  Build container only (about ten minutes).

  python3 tests/golden/make_cfg5_share_facts.py [--work /tmp/cfg5_share]
"""
import argparse
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from strainer2_amd import cfg5  # noqa: E402

EXE = os.path.join(REPO, "oracle", "_ref", "strain_detect")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/cfg5_share")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--prefix-reads", type=int, default=cfg5.PREFIX_READS)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "cfg5_share_facts.json"))
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    assert os.access(EXE, os.X_OK), "build oracle/_ref first (make -C oracle)"
    d = args.work
    t0 = time.time()
    cfg5.write_all(d, procs=args.procs, prefix_reads=args.prefix_reads, only_prefix=True)
    print(f"inputs written in {time.time() - t0:.0f} s", flush=True)
    t1 = time.time()
    ps = []
    for s in cfg5.PINNED_STRAINS:
        cmd = [EXE, "-r", f"s{s}.fa", "-a", f"s{s}.inf", "-b", "prefix.fa", "-t", "SE", "-o", f"ref{s}.gz"]
        ps.append((s, cmd, subprocess.Popen(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    facts = {"workload": "strainer2_amd/cfg5.py: 32 strains x 5 Mbp, 1 %% of the positions informative; SE FASTA of 150 bp reads, 2 %% cut from the "
                         "strains; the first %d reads (%.2f Gbase)" % (args.prefix_reads, args.prefix_reads * cfg5.READ_LEN / 1e9),
             "producer": "oracle/_ref/strain_detect (unmodified reference), one process per pinned strain",
             "prefix_reads": args.prefix_reads, "strains": {}}
    for s, cmd, p in ps:
        out, err = p.communicate()
        assert p.returncode == 0, err
        hits = gzip.open(os.path.join(d, f"ref{s}.gz"), "rb").read()
        facts["strains"][str(s)] = {"argv": cmd[1:-1] + ["<out>"], "returncode": p.returncode, "stdout": out.decode(), "stderr": err.decode(),
                                    "hits_md5": hashlib.md5(hits).hexdigest(), "hits_bytes": len(hits), "hits_lines": hits.count(b"\n"),
                                    "trailer": [ln.decode() for ln in hits.split(b"\n") if ln.startswith(b"#")]}
        print(s, facts["strains"][str(s)]["hits_md5"], facts["strains"][str(s)]["hits_lines"], flush=True)
    print(f"reference: {time.time() - t1:.0f} s", flush=True)
    with open(args.out, "w") as f:
        json.dump(facts, f, indent=1)
    if not args.keep:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
