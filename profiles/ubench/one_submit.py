import sys, numpy as np, torch
sys.path.insert(0, '.')
from tfrec_amd import synth, api
mask = int(sys.argv[1], 0); nb = int(sys.argv[2]); ns = int(sys.argv[3])
host = synth.gen_batch(1000, 0, ns, nb)
r = api.Receiver(n_streams=ns, types_mask=mask, thresh=500, max_blocks=nb, max_events=1 << 16, all_flushes=True)
d = torch.from_numpy(host).cuda()
r.submit(d)
ev = r.drain()
print("mask", hex(mask), "events", len(ev), r.stats())
