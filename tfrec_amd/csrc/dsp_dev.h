// tfrec_amd/csrc/dsp_dev.h -- device-side arithmetic shared by the kernels (exact restatements, see chains.hip).
#pragma once

#include "tfrec_dev.h"
#include "fm_resolve.h"

namespace tfrec {

// (int)double with x86 cvttsd2si semantics (NaN / out of range -> INT_MIN), SURVEY App. E.5
__device__ __forceinline__ int d2i(double v)
{
	if (!(v > -2147483649.0 && v < 2147483648.0))
		return (int)0x80000000;
	return (int)v;
}

// iir2::step (dsp_stuff.cpp:47-56) in the evaluation order of the reference's normative build:
// ((b2*dn2 + a1*yn1) + (b0*dn + b1*dn1)) + a2*yn2, every product and sum rounded (no contraction).
__device__ __forceinline__ double iir_step(Biquad &f, const BiquadCoef &c, double dn)
{
	const double y1 = f.yn, y2 = f.yn1;
	const double y = ((c.b2 * f.dn2 + c.a1 * y1) + (c.b0 * dn + c.b1 * f.dn1)) + c.a2 * y2;
	f.yn1 = y1;
	f.yn = y;
	f.dn2 = f.dn1;
	f.dn1 = dn;
	return y;
}

// The same step with fewer fp64 operations, bit-identical.  iir2::set always yields b2 == b0 and b1 == b0 + b0,
// so with t(n) = fl(b0 * dn(n)):   fl(b2 * dn2) = t(n-2),   fl(b1 * dn1) = 2 * t(n-1) (exact scaling), and
// fl(b0*dn + b1*dn1) = fl(t(n) + 2*t(n-1)) = fma(2, t(n-1), t(n)) (2*t(n-1) is exact, so one rounding).
// Carrying t1 = t(n-1), t2 = t(n-2) next to the state turns 5 mul + 4 add into 3 mul + 1 fma + 3 add.
struct BiquadT {
	double t1, t2;
};
__device__ __forceinline__ BiquadT iirt_enter(const Biquad &f, const BiquadCoef &c)
{
	BiquadT t;
	t.t1 = c.b0 * f.dn1;
	t.t2 = c.b0 * f.dn2;
	return t;
}
__device__ __forceinline__ double iir_step_t(Biquad &f, BiquadT &t, const BiquadCoef &c, double dn)
{
	const double t0 = c.b0 * dn;
	const double s2 = __builtin_fma(2.0, t.t1, t0);
	const double y = ((t.t2 + c.a1 * f.yn) + s2) + c.a2 * f.yn1;
	f.yn1 = f.yn;
	f.yn = y;
	f.dn2 = f.dn1;
	f.dn1 = dn;
	t.t2 = t.t1;
	t.t1 = t0;
	return y;
}

// fm_dev_nrzs, dsp_stuff.cpp:269-279
__device__ __forceinline__ int fm_dev_nrzs(int ar, int aj, int br, int bj)
{
	int cr = (int)((uint32_t)(ar * br) + (uint32_t)(aj * bj));
	cr = cr > 1000000000 ? 1000000000 : cr;
	cr = cr < -1000000000 ? -1000000000 : cr;
	return cr;
}

// The scaled angle atan2(cj, cr) * (16384 / pi) for the generic directions of fm_dev (cj, cr nonzero integers below 2^32
// in magnitude, |cj| != |cr|), absolute error below 2e-11: fm_dev only needs its integer part, and a run certifies itself
// by counting the samples whose scaled angle lies within 1e-9 of an integer (below).  One reciprocal instead of the
// library routine's full division and wide argument handling -- a third of its instructions (the discriminator pass was
// 13 % of the batch's VALU work).  Octant reduction with exact sums of integers, the angle below pi/8 from
// atan(q) = q * P(q^2), degree 10 (Chebyshev fit on [0, tan^2(pi/8)], 3.3e-16), with 16384/pi folded into P's coefficients
// and the reflections done in scaled units, where pi/4, pi/2 and pi are the exact integers 4096, 8192 and 16384 (round 5;
// before, the angle was assembled in radians and scaled by a last multiplication).  Measured against 80-bit references
// on 4e7 random and near-degenerate inputs: |error of the scaled angle| < 6e-12, no integer mismatch (max over the run,
// oracle/mint_golden.py style check in profiles/ubench/atan_check.py).
// The polynomial's coefficients.  Every translation unit that calls fm_dev defines ONE plain (non-const, external)
// __constant__ array from this list and passes it in: the kernel then reads them into scalar registers, and each rides
// along as the scalar operand of its v_fma_f64.  (As literals, or from a const / static array the compiler folds, every
// coefficient costs two v_mov per use: 114 instead of 95 VALU instructions per sample in fmdev_kernel.)
#define TFREC_ATAN_POLY                                                                                                \
	{ 0x1.45f306dc9c882p+12, -0x1.b2995e7b7b081p+10, 0x1.04c26be35c182p+10, -0x1.748375510427cp+9,                        \
	  0x1.21bb891314681p+9, -0x1.da194d380517ep+8, 0x1.91035b898ba13p+8, -0x1.5a01bce7521ebp+8, 0x1.2712f5dc022c6p+8,      \
	  -0x1.bc28ee7da8cf2p+7, 0x1.a1931f2ab065cp+6,                                                                      \
	  /* [0..10]: the radian fit's coefficients times kFmScale = 16384.0 * (1.0 / pi), the reference's factor,            \
	     rounded once (mpmath).  [11..15]: tan(pi/8), 4096, 8192, 16384, kFmScale -- in the same array so that they sit \
	     in scalar registers for the whole kernel instead of being re-materialised (two s_mov each) at every use */    \
	  0x1.a827999fcef32p-2, 4096.0, 8192.0, 16384.0, 16384.0 * (1.0 / 0x1.921fb54442d18p+1) }

// The octant reduction, apart: num == 0 exactly iff the direction is one of the exactly representable ones
// (an axis: mn = 0; a diagonal: mx - mn = 0 in the upper half-octant; the origin: mx = 0) -- ONE compare where fm_dev_fast
// used to test cj == 0, cr == 0 and |cj| == |cr| by themselves (five compares per sample in the discriminator pass).
struct AtanRed {
	double num, den;
	bool upper;
};
__device__ __forceinline__ AtanRed atan2_reduce(double cj, double cr, const double *__restrict__ poly)
{
	const double kTanPi8 = poly[11];
	const double ax = fabs(cr), ay = fabs(cj);
	const double mx = fmax(ax, ay), mn = fmin(ax, ay);
	AtanRed r;
	r.upper = mn > kTanPi8 * mx;  // the angle of (mx, mn) is above pi/8: atan(t) = pi/4 - atan((1-t)/(1+t))
	r.num = r.upper ? mx - mn : mn;
	r.den = r.upper ? mx + mn : mx;  // exact: integers below 2^33
	return r;
}
// num / den: v_rcp_f64 is good to ~2^-23; ONE Newton step (2^-46), the quotient, and one correction of the quotient with
// its exact residual (error 2^-46 * 2^-46: the quotient is then good to an ulp, as after the second Newton step this
// routine made until round 5 -- two instructions less)
__device__ __forceinline__ double atan2_reduced(const AtanRed &r, double cj, double cr, const double *__restrict__ poly)
{
	const double k4096 = poly[12], k8192 = poly[13], k16384 = poly[14];
	const double num = r.num, den = r.den;
	double y = __builtin_amdgcn_rcp(den);
	const double e = __builtin_fma(-den, y, 1.0);
	y = __builtin_fma(y, e, y);
	double q = num * y;
	q = __builtin_fma(__builtin_fma(-den, q, num), y, q);
	const double s2 = q * q;
	double p = poly[10];
#pragma unroll
	for (int k = 9; k >= 0; k--)
		p = __builtin_fma(p, s2, poly[k]);
	double t = q * p;  // scaled: in [0, 2048]
	t = r.upper ? k4096 - t : t;
	t = fabs(cj) > fabs(cr) ? k8192 - t : t;
	t = cr < 0.0 ? k16384 - t : t;
	return copysign(t, cj);
}

// fm_dev, dsp_stuff.cpp:284-292, in the arithmetic of the normative build: (int)(atan2(cj,cr) * (16384/pi)).
// Exactly representable directions (axes, diagonals, signed zeros) are resolved explicitly with the values
// glibc returns for them (times the reference's factor, rounded as the reference's product is) so they do not depend
// on an approximation's last bits; everywhere else the error of atan2_reduced (6e-12 in the scaled angle) can only matter
// when the product is within ~1e-11 of an integer: every sample within 1e-9 is handed to the exact slow path
// (fm_resolve.h), which decides the truncation as the reference does under a correctly rounded atan2 and logs the sample
// for the host's libm check at drain time.
// The exact directions, or a generic one that shares a wave-uniform branch with them: the scaled angle, *generic = false
// for an exact direction.
__device__ __forceinline__ double fm_dev_special(double cr, double cj, const AtanRed &red, bool *generic,
						 const double *__restrict__ atan_poly)
{
	const double kPi = 0x1.921fb54442d18p+1, kPi2 = 0x1.921fb54442d18p+0, kPi4 = 0x1.921fb54442d18p-1,
		     k3Pi4 = 0x1.2d97c7f3321d2p+1;
	const double kFmScale = atan_poly[15];
	*generic = false;
	if (cj == 0.0) {
		const bool pos = cr > 0.0 || (cr == 0.0 && !signbit(cr));
		return copysign(pos ? 0.0 : kPi, cj) * kFmScale;
	}
	if (cr == 0.0)
		return copysign(kPi2, cj) * kFmScale;
	if (fabs(cj) == fabs(cr))
		return copysign(cr > 0.0 ? kPi4 : k3Pi4, cj) * kFmScale;
	*generic = true;
	return atan2_reduced(red, cj, cr, atan_poly);
}
// Returns the scaled angle; true = generic direction within 1e-9 of a truncation boundary.
__device__ __forceinline__ bool fm_dev_fast(double cr, double cj, double *v_out, const double *__restrict__ atan_poly,
					    double flag_eps = 1e-9)
{
	double v;
	bool generic = true;
	// The exactly representable directions are a handful of samples per batch: ONE wave-uniform test keeps the four-way
	// divergent chain of cases (a dozen exec-mask instructions per sample, a third of the discriminator pass's scalar
	// instructions) out of the samples' common path.
	const AtanRed red = atan2_reduce(cj, cr, atan_poly);
	const bool special = red.num == 0.0;  // cj == 0 || cr == 0 || |cj| == |cr| (see atan2_reduce)
	if (__builtin_expect(__ballot(special) != 0ull, 0))
		v = fm_dev_special(cr, cj, red, &generic, atan_poly);
	else
		v = atan2_reduced(red, cj, cr, atan_poly);
	*v_out = v;
	return generic && fabs(v - rint(v)) < flag_eps;
}

// double-double sin / cos tables of the slow path (8 KB, touched ~once per 5e8 samples); one copy per translation unit
static __device__ const double kFmCoarse[129][4] = { TFREC_FM_COARSE_TABLE };
static __device__ const double kFmFine[128][4] = { TFREC_FM_FINE_TABLE };

// The exact decision for one flagged sample + its log entry.  Not inlined: the fp64 double-double code would otherwise
// set the register budget of the kernels that call fm_dev for a branch taken once per ~5e8 samples.
__device__ __noinline__ int fm_dev_slow(double cr, double cj, double v_fast, EventBuf *eb)
{
	double margin;
	const int r = fm_dev_resolve(cr, cj, v_fast, kFmCoarse, kFmFine, &margin);
	atomicAdd(&eb->uncertain, 1ull);
	if (margin < kFmUndecidableUlps)
		atomicAdd(&eb->fm_undecidable, 1u);
	const uint32_t i = atomicAdd(&eb->fm_logged, 1u);
	if (i < (uint32_t)kFmLogCap) {
		eb->fm_log[i].cr = cr;
		eb->fm_log[i].cj = cj;
		eb->fm_log[i].result = r;
		eb->fm_log[i].margin = (float)margin;
	}
	return r;
}

// cr, cj: the cross terms of dsp_stuff.cpp:288-289 (exact integers)
__device__ __forceinline__ int fm_dev_cross(double cr, double cj, EventBuf *eb, const double *__restrict__ atan_poly)
{
	double v;
	const bool unc = fm_dev_fast(cr, cj, &v, atan_poly);
	int r = d2i(v);
	if (__builtin_expect(unc, 0))
		r = fm_dev_slow(cr, cj, v, eb);
	return r;
}

__device__ __forceinline__ int fm_dev(int ar, int aj, int br, int bj, EventBuf *eb, const double *__restrict__ atan_poly)
{
	const double cr = ((double)ar) * br + ((double)aj) * bj;
	const double cj = ((double)aj) * br - ((double)ar) * bj;
	return fm_dev_cross(cr, cj, eb, atan_poly);
}

}  // namespace tfrec
