"""CPU check of the atan2 approximation in tfrec_amd/csrc/dsp_dev.h (atan2_int): same operation sequence in float64
(fused multiply-adds emulated in 80-bit), against libm atan2 and an 80-bit reference.  python atan_check.py"""
import numpy as np
ld = np.longdouble
C = [float.fromhex(h) for h in ['0x1.fffffffffffffp-1', '-0x1.5555555555101p-2', '0x1.9999999915220p-3', '-0x1.249248f459b71p-3', '0x1.c71c601c68b53p-4', '-0x1.745b3a024febep-4', '0x1.3af4788c30195p-4', '-0x1.0fc0caec4e264p-4', '0x1.cf80524e56f02p-5', '-0x1.5cd7a4fac9dc7p-5', '0x1.47f65fb716232p-6']]
kPi = float.fromhex('0x1.921fb54442d18p+1'); kPi2 = kPi/2; kPi4 = kPi/4
kTan = float(np.tan(ld(np.pi)/8))
kScale = 16384.0 * (1.0 / kPi)
def fma(a, b, c):
    return (a.astype(ld) * b.astype(ld) + np.asarray(c, dtype=ld)).astype(np.float64)
def fast(cj, cr):
    ax, ay = np.abs(cr), np.abs(cj)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    upper = mn > kTan * mx
    num = np.where(upper, mx - mn, mn); den = np.where(upper, mx + mn, mx)
    y = (1.0 / den).astype(np.float32).astype(np.float64)   # a crude rcp, like v_rcp_f64's worst case
    e = fma(-den, y, 1.0); y = fma(y, e, y)
    e = fma(-den, y, 1.0); y = fma(y, e, y)
    q = num * y
    rr = fma(-den, q, num); q = fma(rr, y, q)
    s2 = q * q
    p = np.full_like(q, C[10])
    for c in C[9::-1]:
        p = fma(p, s2, c)
    phi = q * p
    phi = np.where(upper, kPi4 - phi, phi)
    phi = np.where(ay > ax, kPi2 - phi, phi)
    phi = np.where(cr < 0, kPi - phi, phi)
    return np.copysign(phi, cj)
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
    v = fast(cj, cr) * kScale
    truth = (np.arctan2(cj.astype(ld), cr.astype(ld)) * (ld(16384) / ld(np.pi)))
    worst = max(worst, float(np.max(np.abs(v.astype(ld) - truth))))
    bad += int(np.sum(np.trunc(v) != np.trunc(v_ref)))
    near += int(np.sum(np.abs(v_ref - np.rint(v_ref)) < 1e-9))
    n_tot += len(cr)
print("samples", n_tot, "integer mismatches", bad, "max |v - true v|", worst, "reference within 1e-9 of an integer:", near)
