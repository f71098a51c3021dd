"""whb_demod_kernel by itself: a WHB-only context (types 0x20), one submit at a time, HIP-event time of the kernel; with
a library built with -DTFREC_AMD_PROFILE_WHB also its cycle counters.  usage: [TFREC_AMD_LIB=...] whb_alone.py [n]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tfrec_amd import synth, api
ns, nb = 1024, 48
host = synth.gen_batch(1000, 0, ns, nb)
d = torch.from_numpy(host).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with api.Receiver(ns, 0x20, 500, 0, max_blocks=nb, max_events=1 << 20, timing=True) as r:
    t = []
    for _ in range(n):
        r.submit(d); r.drain()
        t.append(r.timings())
    st = api.Stats()
    r.L.tfrec_amd_get_stats(r.h, C.byref(st))
    raw = [int(x) for x in (st.tfa1_recomputed, *st.reserved)]
    parts = [int(st.biquad_unconverged), int(st.biquad_serial), int(st.tfa2_resliced)]
print("whb_demod_ms", ["%.2f" % x["whb_demod_ms"] for x in t], "whb_biquad_ms %.2f frontend %.2f" % (t[-1]["whb_biquad_ms"], t[-1]["frontend_ms"]))
steps, usteps = raw[0] >> 32, raw[0] & 0xffffffff
if steps:
    print("per stream and submit: steps %.0f, with recurrence %.0f; cycles: recurrence %.2fM (%.1f per sample), whole %.2fM" % (
        steps / ns / n, usteps / ns / n, raw[1] / ns / n / 1e6, raw[1] / max(usteps, 1) / 64, raw[3] / ns / n / 1e6))
    print("  step top (synced steps: top + mask) %.2fM, candidate walk %.2fM, step tail %.2fM cycles per stream and submit" % tuple(x / ns / n / 1e6 for x in parts))
