#!/usr/bin/env python3
"""Full-size parity with other data than the bench's and the tests': 1024 streams x 48 blocks x five protocols, every flush of every
stream against the oracle, two submits of 24 blocks (windows cut), for the seeds / noise levels on the command line.
usage: python profiles/ubench/fullsize_check.py <seed>[:noise_q8] ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from tfrec_amd import api, synth  # noqa: E402
from test_gpu_parity import _all_streams_equal  # noqa: E402

bad = 0
for arg in sys.argv[1:]:
    seed, _, noise = arg.partition(":")
    seed, noise = int(seed), int(noise or 256)
    n_streams, n_blocks, cut = 1024, 48, 24
    t0 = time.time()
    iq = synth.gen_batch(seed, 0, n_streams, n_blocks, 0x1F, noise)
    with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=cut, all_flushes=True, max_events=1 << 20) as r:
        evs = []
        for k in range(n_blocks // cut):
            r.submit(np.ascontiguousarray(iq[:, k * cut * 65536:(k + 1) * cut * 65536]))
        for k in range(n_blocks // cut):
            evs.append(r.drain())
        st = r.stats()
    ev = np.concatenate(evs)
    try:
        n = _all_streams_equal(ev, iq, 0x2F, 500)
        print("seed %d noise %d: %d flush events of %d streams equal the oracle's (%.0f s); scalar groups TFA_1 %d, TFA_2 family %d" % (
            seed, noise, n, n_streams, time.time() - t0, st["tfa1_scalar_groups"], st["tfa2_scalar_groups"]), flush=True)
    except AssertionError as e:
        bad += 1
        print("seed %d noise %d: MISMATCH %s" % (seed, noise, e), flush=True)
sys.exit(1 if bad else 0)
