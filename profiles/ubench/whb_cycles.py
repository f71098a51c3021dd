"""Cycle accounting of whb_demod_kernel (library built with -DTFREC_AMD_PROFILE_WHB[=1]: the two printed parts are
the recurrence and the candidate walk; =2: loop top -> recurrence, after the candidates -> loop end; =3: after the
candidates -> state update, state update -> loop end).  usage: TFREC_AMD_LIB=.../lib_prof.so TFREC_AMD_WHB2=0 python whb_cycles.py [types_hex]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from tfrec_amd import synth, api
mask = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0x20
ns, nb = 1024, 48
host = synth.gen_batch(1000, 0, ns, nb)
d = torch.from_numpy(host).cuda()
with api.Receiver(ns, mask, 500, 0, max_blocks=nb, max_events=1 << 20) as r:
    pipelined = len(sys.argv) > 2 and sys.argv[2] == "pipelined"
    if pipelined:  # three submits in flight, as bench.py runs: the front end of later batches beside this kernel
        r.submit(d); r.submit(d)
        for _ in range(6):
            r.submit(d); r.drain()
        r.drain(); r.drain()
        nsub = 8
    else:
        nsub = 2
        for _ in range(2):
            r.submit(d); r.drain()
    st = api.Stats()
    r.L.tfrec_amd_get_stats(r.h, C.byref(st))
    raw = [int(x) for x in (st.tfa1_recomputed, st.biquad_repair_slots, st.whb_respeculated, st.tfa1_scalar_groups)]
    span = [int(st.biquad_unconverged), int(st.biquad_serial), int(st.tfa2_resliced)]
steps, usteps = raw[0] >> 32, raw[0] & 0xffffffff
print("per stream and submit: steps %.0f, with recurrence %.0f" % (steps / ns / nsub, usteps / ns / nsub))
print("cycles per stream and submit: recurrence %.2fM, whole demodulator %.2fM; wall %.3f ms per stream -> shader clock %.2f GHz" % (
    raw[1] / ns / nsub / 1e6, raw[3] / ns / nsub / 1e6, raw[2] / ns / nsub / 1e5, raw[3] / max(raw[2], 1) / 10.0))
print("recurrence: %.0f cycles per step = %.1f per sample" % (raw[1] / max(usteps, 1), raw[1] / max(usteps, 1) / 64))
if not (len(sys.argv) > 3 and sys.argv[3] == "span"):
    print("cycles per step (all steps): loop top -> recurrence %.0f, recurrence (steps with it) %.0f, candidates %.0f, state update + loop end %.0f; whole kernel / steps %.0f" % (
        span[0] / max(steps, 1), raw[1] / max(usteps, 1), span[1] / max(steps, 1), span[2] / max(steps, 1), raw[3] / max(steps, 1)))
if len(sys.argv) > 3 and sys.argv[3] == "span":  # library built with -DTFREC_AMD_PROFILE_WHB_SPAN as well
    first = (~span[0]) & 0xFFFFFFFFFFFFFFFF
    mean = (int(st.biquad_segments) - ns * (first & 0xffffffffff)) / ns  # (low 40 bits summed)
    print("sixth submit: workgroups start over %.3f ms (mean start +%.3f ms), kernel first start -> last end %.3f ms" % (
        (span[1] - first) / 1e5, mean / 1e5, (span[2] - first) / 1e5))
    print("slowest stream of that submit: %.2fM cycles, %d steps" % ((raw[1] >> 24) / 1e6, raw[1] & 0xffffff))
    print("streams by cycles per step: <2500: %d, <3000: %d, <3500: %d, <4500: %d, more: %d" % tuple((raw[2] >> (12 * b)) & 0xfff for b in range(5)))
