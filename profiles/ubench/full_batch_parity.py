"""The bench batch (1024 distinct streams x 48 blocks, all protocols), five batches through the four-deep FIFO: every flush event of the
window-parallel pipeline must equal the one of the independent serial lane-per-chain GPU implementation."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tfrec_amd import api, synth

ns, nb = 1024, 48
d = [torch.from_numpy(synth.gen_batch(1000 + 17 * k, 0, ns, nb)).cuda() for k in range(5)]
out = {}
for serial in (False, True):
    with api.Receiver(ns, 0x2F, 500, 0, max_blocks=nb, all_flushes=True, serial_chains=serial, max_events=ns * 400) as r:
        if serial:
            evs = []
            for x in d:
                r.submit(x)
                evs.append(r.drain())
        else:
            evs = []
            for k, x in enumerate(d):
                if k >= api.FIFO_DEPTH:
                    evs.append(r.drain())
                r.submit(x)
            while len(evs) < len(d):
                evs.append(r.drain())
        assert r.atan_uncertain() == 0
        out[serial] = evs
for k in range(len(d)):
    a, b = out[False][k], out[True][k]
    print("batch %d: %d events, equal: %s" % (k, len(a), a.tobytes() == b.tobytes()))
    assert a.tobytes() == b.tobytes()
print("ok")
