#!/usr/bin/env python3
"""fp32 pre-estimate of fm_dev's scaled angle (round 6): fit of atan(q) / q = P(q^2) on [0, tan(pi/8)] with 16384 / pi folded
in, and the error of the whole fp32 evaluation (as fmdev_kernel does it, emulated with numpy float32) against fp64 on random
and adversarial cross terms.  Prints the coefficients as hex floats and the maximum |error| in scaled-angle units.
usage: fm32_fit.py [degree=4] [n=4000000]"""
import sys
import numpy as np

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
K = 16384.0 / np.pi
T8 = np.tan(np.pi / 8)
# minimax-ish: least squares on Chebyshev nodes of s = q^2 in [0, T8^2 * 1.02], relative weight
m = 4000
x = np.cos(np.pi * (np.arange(m) + 0.5) / m)
s = (x + 1) / 2 * (T8 * 1.01) ** 2
q = np.sqrt(s)
f = np.where(q > 0, np.arctan(q) / np.where(q > 0, q, 1), 1.0) * K
A = np.vander(s, deg + 1, increasing=True)
c, *_ = np.linalg.lstsq(A, f, rcond=None)
# a few Remez-like reweighting rounds
w = np.ones(m)
for _ in range(40):
    r = A @ c - f
    w *= (1 + 4 * np.abs(r) / np.abs(r).max())
    w /= w.mean()
    c, *_ = np.linalg.lstsq(A * w[:, None], f * w, rcond=None)
r = A @ c - f
print("degree", deg, "fit error of P (scaled units, times q <= 0.42):", np.abs(r).max(), "-> angle", np.abs(r * q).max())
c32 = c.astype(np.float32)
print("coefficients (float32):", ", ".join(float(v).hex() for v in c32))
print("as decimals:", ", ".join(repr(float(v)) for v in c32))

rng = np.random.default_rng(1)
def sample(n):
    # decimated samples of u8 input never leave +-12153; also full-range int16 pairs
    lim = rng.choice([12153, 32767, 300, 40], size=n, p=[0.5, 0.2, 0.2, 0.1])
    I = (rng.random(n) * 2 - 1) * lim; Q = (rng.random(n) * 2 - 1) * lim
    pI = (rng.random(n) * 2 - 1) * lim; pQ = (rng.random(n) * 2 - 1) * lim
    I, Q, pI, pQ = [np.rint(v).astype(np.int64) for v in (I, Q, pI, pQ)]
    return I * pI + Q * pQ, Q * pI - I * pQ
cr, cj = sample(n)
# adversarial: near the octant boundaries and near tan(pi/8)
k = n // 8
base = rng.integers(1, 1 << 28, size=k)
adv = [(base, base + rng.integers(-3, 4, size=k)), (base, np.rint(base * T8).astype(np.int64) + rng.integers(-3, 4, size=k)),
       (base, rng.integers(-3, 4, size=k)), (-base, base + rng.integers(-3, 4, size=k))]
cr = np.concatenate([cr] + [a for a, b in adv]); cj = np.concatenate([cj] + [b for a, b in adv])
ok = (cr != 0) | (cj != 0)
cr, cj = cr[ok], cj[ok]
ref = np.abs(np.arctan2(cj.astype(np.float64), cr.astype(np.float64))) * K   # magnitude of the scaled angle
F = np.float32
ax = np.abs(cr).astype(F); ay = np.abs(cj).astype(F)     # v_cvt_f32_i32 (RNE) + abs
mx = np.maximum(ax, ay); mn = np.minimum(ax, ay)
upper = mn > F(T8) * mx
num = np.where(upper, mx - mn, mn).astype(F); den = np.where(upper, mx + mn, mx).astype(F)
y = (F(1) / den).astype(F)
# v_rcp_f32 is good to 1 ulp: perturb by one ulp either way at random to cover it
y = np.nextafter(y, np.where(rng.random(y.size) < 0.5, F(0), F(np.inf)).astype(F)).astype(F)
qq = (num * y).astype(F)
s2 = (qq * qq).astype(F)
p = np.full(qq.shape, c32[deg], dtype=F)
for kk in range(deg - 1, -1, -1):
    p = (p * s2 + c32[kk]).astype(F)     # (fma: one rounding; numpy rounds twice -- the bound below has the margin)
t = (qq * p).astype(F)
t = np.where(upper, F(4096) - t, t).astype(F)
t = np.where(ay > ax, F(8192) - t, t).astype(F)
t = np.where(cr < 0, F(16384) - t, t).astype(F)
err = np.abs(t.astype(np.float64) - ref)
print("samples", err.size, "max |fp32 - fp64| of the scaled angle:", err.max(), " 99.99 %:", np.quantile(err, 0.9999))
for thr in (1 / 64, 1 / 128, 1 / 256):
    dist = np.abs(t - np.rint(t))
    dec = dist >= thr
    wrong = dec & (np.trunc(t.astype(np.float64)) != np.trunc(ref))
    print("threshold 1/%d: decided %.4f of the samples, wrong decisions %d" % (round(1 / thr), dec.mean(), wrong.sum()))
