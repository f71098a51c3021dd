"""Is the batch bound by the chip's throughput or by the length of its kernel chains?  The same 1024 streams as ONE
context (the bench) and as K contexts of 1024/K streams on the same device, each with its own streams and FIFO, driven
round-robin from one host thread.  usage: python profiles/ubench/two_contexts.py [K ...]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from tfrec_amd import api, synth

N, B, TYPES, STEPS, WARM = 1024, 48, 0x2F, 30, 6
dev = torch.device("cuda:0")
uniq = 64
host = np.stack([synth.gen_stream(1000, s, B, 0x1F, 256) for s in range(uniq)])
d_u = torch.from_numpy(host).to(dev)
for K in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    n = N // K
    d_iq = torch.empty((n, host.shape[1]), dtype=torch.uint8, device=dev)
    for s0 in range(0, n, uniq):
        k = min(uniq, n - s0)
        d_iq[s0:s0 + k].copy_(d_u[:k])
    torch.cuda.synchronize()
    rs = [api.Receiver(n, TYPES, 500, 0, device=0, max_blocks=B, max_events=max(4096, n * 256)) for _ in range(K)]
    depth = api.FIFO_DEPTH

    def run(steps):
        q = 0
        for k in range(steps):
            while q < steps and q - k < depth:
                for r in rs:
                    r.submit(d_iq)
                q += 1
            for r in rs:
                r.drain()
    run(WARM)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(STEPS)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print("%d context(s) x %d streams: %.3f ms per %d streams" % (K, n, dt * 1e3, N), flush=True)
    for r in rs:
        r.close()
