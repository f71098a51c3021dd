import sys, numpy as np, torch
sys.path.insert(0, '.')
from tfrec_amd import synth, api
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 512
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
host = synth.gen_batch(seed, 0, ns, nb)
lo, hi = 0, ns
# bisect on stream ranges
def count(a, b):
    with api.Receiver(b - a, 0x2F, 500, 0, max_blocks=nb) as r:
        r.submit(np.ascontiguousarray(host[a:b])); r.drain()
        r.submit(np.ascontiguousarray(host[a:b])); r.drain()
        return r.stats()["tfa2_resliced"]
print("total", count(0, ns))
found = []
step = 256
for a in range(0, ns, step):
    c = count(a, min(ns, a + step))
    if c:
        for s in range(a, min(ns, a + step)):
            if count(s, s + 1):
                found.append(s)
print("streams with reslice:", found)
