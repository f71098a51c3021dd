import os, sys
sys.path.insert(0, '/root/repo')
os.environ["TFREC_AMD_FM_FLAG_EPS"] = sys.argv[1]
import numpy as np
from tfrec_amd import api, synth
n_streams, n_blocks = 4, 16
iq = synth.gen_batch(31, 7, n_streams, n_blocks)
with api.Receiver(n_streams, 0x2F, 500, 0, max_blocks=n_blocks, all_flushes=True) as r:
    for h in range(2):
        r.submit(np.ascontiguousarray(iq[:, h * (n_blocks // 2) * 65536:(h + 1) * (n_blocks // 2) * 65536]))
    ev = np.concatenate([r.drain(), r.drain()])
    print(sys.argv[1], len(ev), r.fm_stats())
