#!/bin/bash
# tools/probes/gpu_inflate_spec.sh -- run on the GPU box: a FASTQ .gz (READS x 150 bp, binned qualities, zlib level 4) through the
# whole speculative decode on the device (gpu_inflate_spec.hip), chain, length and CRC-32 checked on the host.  SEG_KIB: segment size.
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
for S in ${SEG_KIB:-64 32}; do timeout -k 10 300 tools/probes/gpu_inflate_spec $W/t.fq.gz $S || echo "(exit $?)"; done
rm -rf $W
