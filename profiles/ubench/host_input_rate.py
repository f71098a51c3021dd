"""PCIe-inclusive rate: batches fed from page-locked HOST memory with tfrec_amd_submit_host, two submits in flight
(what the file feeder of tfrec_amd/host/gpu_engine.cpp does).  usage: host_input_rate.py [streams] [blocks] [steps]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tfrec_amd import api, synth

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 48
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
L = api.load_library()
L.tfrec_amd_host_alloc.restype = C.c_void_p
L.tfrec_amd_host_alloc.argtypes = [C.c_size_t]
row = n_blocks * api.BLOCK_BYTES
bufs = []
src = synth.gen_batch(1000, 0, min(n_streams, 64), n_blocks)
for k in range(3):
    p = L.tfrec_amd_host_alloc(n_streams * row)
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n_streams, row))
    for s0 in range(0, n_streams, src.shape[0]):
        a[s0:s0 + src.shape[0]] = src[: min(src.shape[0], n_streams - s0)]
    bufs.append(a)
with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, max_events=n_streams * 256) as r:
    def run(n):
        r.submit(bufs[0])
        for k in range(n):
            if k + 1 < n:
                r.submit(bufs[(k + 1) % 3])
            r.drain()
    run(2)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
print("host-input (PCIe-inclusive): %.1f MSamples/s, %.2f ms per batch, %.1f GB/s over PCIe" % (
    n_streams * n_blocks * 32768 * steps / dt / 1e6, dt / steps * 1e3, n_streams * row * steps / dt / 1e9))
