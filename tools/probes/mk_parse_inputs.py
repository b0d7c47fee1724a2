import numpy as np
rng=np.random.default_rng(1)
n=1_500_000
bases=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,size=(n,150))]
with open('/dev/shm/pp/r.fa','wb') as f, open('/dev/shm/pp/r.fq','wb') as g:
    q=b"I"*150
    for i in range(n):
        s=bases[i].tobytes()
        f.write(b">read%d/1\n"%i+s+b"\n")
        g.write(b"@read%d/1\n"%i+s+b"\n+\n"+q+b"\n")
