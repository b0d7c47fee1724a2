#!/bin/bash
# tools/probes/gpu_inflate_probe.sh -- run on the GPU box: a FASTQ .gz (READS x 150 bp, binned qualities, zlib level 4), its blocks
# dumped by the host decoder, every block decoded on the device one lane per block.  (Build both programs first, see their headers.)
set -e
READS=${READS:-6000000}
W=/dev/shm/sk_gpuinf
mkdir -p $W
python3 - <<PY
import zlib, numpy as np
rng = np.random.default_rng(1)
co = zlib.compressobj(4, zlib.DEFLATED, 31)
with open("$W/t.fq.gz", "wb") as f:
    for a0 in range(0, $READS, 1000000):
        n = min(1000000, $READS - a0)
        out = np.empty((n, 3 + 151 + 2 + 151), dtype=np.uint8)
        out[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        out[:, 3:153] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
        out[:, 153] = 10
        out[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
        out[:, 156:306] = rng.choice(np.frombuffer(b"FFFFFFFFFF::,#", dtype=np.uint8), size=(n, 150))
        out[:, 306] = 10
        f.write(co.compress(out.tobytes()))
    f.write(co.flush())
PY
ls -la $W/t.fq.gz
tools/probes/gz_blocks_dump $W/t.fq.gz $W/t.bin ${BLOCKS:-40000}
timeout -k 10 120 tools/probes/gpu_inflate_probe $W/t.fq.gz $W/t.bin
rm -rf $W
