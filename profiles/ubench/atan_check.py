"""CPU check of the atan2 approximation in tfrec_amd/csrc/dsp_dev.h (atan2_reduce + atan2_reduced; until round 4: atan2_int): same operation sequence in float64
(fused multiply-adds emulated in 80-bit), against libm atan2 and an 80-bit reference.  python atan_check.py"""
import numpy as np
ld = np.longdouble
C = [float.fromhex(h) for h in ['0x1.45f306dc9c882p+12', '-0x1.b2995e7b7b081p+10', '0x1.04c26be35c182p+10', '-0x1.748375510427cp+9', '0x1.21bb891314681p+9', '-0x1.da194d380517ep+8', '0x1.91035b898ba13p+8', '-0x1.5a01bce7521ebp+8', '0x1.2712f5dc022c6p+8', '-0x1.bc28ee7da8cf2p+7', '0x1.a1931f2ab065cp+6']]
kPi = float.fromhex('0x1.921fb54442d18p+1')
kTan = float(np.tan(ld(np.pi)/8))
kScale = 16384.0 * (1.0 / kPi)
def fma(a, b, c):
    return (a.astype(ld) * b.astype(ld) + np.asarray(c, dtype=ld)).astype(np.float64)
def fast(cj, cr):
    """atan2_reduced of dsp_dev.h: the scaled angle (coefficients carry 16384/pi; reflections about 4096, 8192, 16384)"""
    ax, ay = np.abs(cr), np.abs(cj)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    upper = mn > kTan * mx
    num = np.where(upper, mx - mn, mn); den = np.where(upper, mx + mn, mx)
    y = (1.0 / den).astype(np.float32).astype(np.float64)   # a crude rcp, like v_rcp_f64's worst case (2^-24)
    e = fma(-den, y, 1.0); y = fma(y, e, y)                  # ONE Newton step
    q = num * y
    rr = fma(-den, q, num); q = fma(rr, y, q)                # the quotient corrected with its exact residual
    s2 = q * q
    p = np.full_like(q, C[10])
    for c in C[9::-1]:
        p = fma(p, s2, c)
    t = q * p
    t = np.where(upper, 4096.0 - t, t)
    t = np.where(ay > ax, 8192.0 - t, t)
    t = np.where(cr < 0, 16384.0 - t, t)
    return np.copysign(t, cj)
rng = np.random.default_rng(1)
worst = 0; bad = 0; n_tot = 0; near = 0
for it in range(40):
    n = 1_000_000
    if it % 4 == 0:   # small amplitudes (noise)
        a = rng.integers(-300, 300, size=(4, n))
    elif it % 4 == 1:
        a = rng.integers(-32768, 32768, size=(4, n))
    elif it % 4 == 2:
        a = rng.integers(-4000, 4000, size=(4, n))
    else:   # nearly aligned / nearly diagonal directions
        a = rng.integers(-20000, 20000, size=(4, n)); a[2] = a[0] + rng.integers(-3, 4, size=n); a[3] = a[1] + rng.integers(-3, 4, size=n)
    ar, aj, br, bj = a.astype(np.float64)
    cr = ar * br + aj * bj; cj = aj * br - ar * bj
    gen = (cj != 0) & (cr != 0) & (np.abs(cj) != np.abs(cr))
    cr, cj = cr[gen], cj[gen]
    v_ref = np.arctan2(cj, cr) * kScale
    v = fast(cj, cr)
    truth = (np.arctan2(cj.astype(ld), cr.astype(ld)) * ld(kScale))   # the reference's factor, exactly
    worst = max(worst, float(np.max(np.abs(v.astype(ld) - truth))))
    bad += int(np.sum(np.trunc(v) != np.trunc(v_ref)))
    near += int(np.sum(np.abs(v_ref - np.rint(v_ref)) < 1e-9))
    n_tot += len(cr)
print("samples", n_tot, "integer mismatches", bad, "max |v - true v|", worst, "reference within 1e-9 of an integer:", near)
