// tfrec_amd/csrc/dsp_dev.h -- device-side arithmetic shared by the kernels (exact restatements, see chains.hip).
#pragma once

#include "tfrec_dev.h"

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

// fm_dev, dsp_stuff.cpp:284-292, in the arithmetic of the normative build: (int)(atan2(cj,cr) * (16384/pi)).
// Exactly representable directions (axes, diagonals, signed zeros) are resolved explicitly with the values
// glibc returns for them so they do not depend on the device atan2's last bit; everywhere else a 1-2 ulp
// difference can only matter when the product is within ~1e-11 of an integer; such samples are counted
// (threshold 1e-9) so a run can certify itself (DESIGN.md "fm_dev").
__device__ __forceinline__ int fm_dev(int ar, int aj, int br, int bj, bool *uncertain)
{
	const double cr = ((double)ar) * br + ((double)aj) * bj;
	const double cj = ((double)aj) * br - ((double)ar) * bj;
	const double kPi = 0x1.921fb54442d18p+1, kPi2 = 0x1.921fb54442d18p+0, kPi4 = 0x1.921fb54442d18p-1,
		     k3Pi4 = 0x1.2d97c7f3321d2p+1;
	const double kScale = 16384.0 * (1.0 / 0x1.921fb54442d18p+1);
	double ang;
	bool generic = false;
	if (cj == 0.0) {
		const bool pos = cr > 0.0 || (cr == 0.0 && !signbit(cr));
		ang = copysign(pos ? 0.0 : kPi, cj);
	} else if (cr == 0.0) {
		ang = copysign(kPi2, cj);
	} else if (fabs(cj) == fabs(cr)) {
		ang = copysign(cr > 0.0 ? kPi4 : k3Pi4, cj);
	} else {
		ang = atan2(cj, cr);
		generic = true;
	}
	const double v = ang * kScale;
	*uncertain = generic && fabs(v - rint(v)) < 1e-9;
	return d2i(v);
}

}  // namespace tfrec
