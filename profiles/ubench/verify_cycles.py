"""whb_verify_kernel: samples per stream and cycles (library built with -DTFREC_AMD_PROFILE_VERIFY, which replaces the
speculation counters by: samples sum / max per lane, wave cycles sum / max, iterations).
usage: TFREC_AMD_LIB=tfrec_amd/ab/pv.so python profiles/ubench/verify_cycles.py [types_hex]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from tfrec_amd import synth, api
mask = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0x20
ns, nb = 1024, 48
host = synth.gen_batch(1000, 0, ns, nb)
d = torch.from_numpy(host).cuda()
with api.Receiver(ns, mask, 500, 0, max_blocks=nb, max_events=1 << 20, timing=True) as r:
    r.submit(d); r.drain()
    st = api.Stats()
    r.L.tfrec_amd_get_stats(r.h, C.byref(st))
    raw = [int(st.biquad_segments), int(st.biquad_unconverged), int(st.biquad_serial), int(st.tfa2_resliced), int(st.tfa1_recomputed)]
    t = r.timings()
print("filter samples per stream: mean %.0f, max %d; half-steps per stream mean %.0f" % (raw[0] / ns, raw[1], raw[4] / ns))
print("verify kernel: %.3f ms (HIP events); slowest wave %.2fM cycles = %.1f cycles per sample of the slowest stream; mean wave %.2fM cycles"
      % (t["whb_verify_ms"], raw[3] / 1e6, raw[3] / max(raw[1], 1), raw[2] / ns / 1e6))
print("whb_demod %.3f ms" % t["whb_demod_ms"])
