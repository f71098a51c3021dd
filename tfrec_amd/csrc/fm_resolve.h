// tfrec_amd/csrc/fm_resolve.h -- exact slow path of fm_dev (dsp_stuff.cpp:284-292) for the samples the fast path
// cannot certify.  Compiles for the device (hipcc) and for the host (g++: tests/test_fm_resolve_cpu.py drives it
// against libm); plain IEEE fp64 with explicit fma, no contraction (-ffp-contract=off on both sides).
//
// The reference computes  r = (int)( atan2(cj, cr) * K ),  K = 16384 * fl(1/pi),  in double.  For generic directions
// (cr, cj nonzero integers below 2^32, |cr| != |cj|) the fast path (dsp_dev.h: atan2_reduce + atan2_reduced) knows the scaled angle to
// 6e-12; when that is within 1e-9 of an integer k the truncation is decided HERE, exactly:
//
//   1. a_k := the smallest double a with fl(a * K) >= k (found by stepping from fl(k / K)): the reference returns
//      +-k iff its atan2 result R satisfies R >= a_k, else +-(k - 1).
//   2. A correctly rounded atan2 gives R >= a_k iff theta > m, m := the midpoint of pred(a_k) and a_k (theta = m is
//      impossible: tan of a nonzero dyadic rational is transcendental, cj/cr is rational).
//   3. theta > m  iff  sin(theta - m) > 0  iff  D := |cj| cos m - cr sin m > 0.  With phi = k pi / 16384 and
//      eps = m - phi (|eps| < 3e-15), sin / cos(phi) come from two double-double tables (k = 128 i + j, angle
//      addition; fm_resolve_tables.h), and D = (|cj| C - cr S) - eps (|cj| S + cr C) is evaluated with exact products
//      (fma) and a double-double difference: error below 1e-30 |c|, i.e. theta - m is resolved to ~1e-30 rad.
//
// What this cannot know: glibc's atan2 is not correctly rounded (<= 0.55 ulp; and its x86-64 ifunc picks an FMA or a
// non-FMA build by CPU), so for theta within 0.05 ulp of m the reference binary itself may go either way.  Such
// samples (expected ~1e-13 per sample) are reported as `undecidable`; the caller logs every slow-path sample and the
// host checks it against ITS libm when the batch is drained (capi.hip), so a run certifies itself.
#pragma once

#include <math.h>
#include <stdint.h>

#include "fm_resolve_tables.h"

#if defined(__HIPCC__)
#define TFREC_HD __host__ __device__ __forceinline__
#else
#define TFREC_HD static inline
#endif

namespace tfrec {

struct FmDD {
	double hi, lo;
};

TFREC_HD FmDD fm_two_sum(double a, double b)
{
	const double s = a + b;
	const double bb = s - a;
	return FmDD{ s, (a - (s - bb)) + (b - bb) };
}
TFREC_HD FmDD fm_fast_two_sum(double a, double b)  // |a| >= |b|
{
	const double s = a + b;
	return FmDD{ s, b - (s - a) };
}
TFREC_HD FmDD fm_dd_mul(FmDD a, FmDD b)
{
	const double p = a.hi * b.hi;
	double e = __builtin_fma(a.hi, b.hi, -p);
	e += a.hi * b.lo + a.lo * b.hi;
	return fm_fast_two_sum(p, e);
}
TFREC_HD FmDD fm_dd_add(FmDD a, FmDD b)
{
	FmDD s = fm_two_sum(a.hi, b.hi);
	const FmDD t = fm_two_sum(a.lo, b.lo);
	s.lo += t.hi;
	s = fm_fast_two_sum(s.hi, s.lo);
	s.lo += t.lo;
	return fm_fast_two_sum(s.hi, s.lo);
}
TFREC_HD double fm_bits_step(double a, int dir)  // next double above (dir > 0) / below a positive finite a
{
	uint64_t u;
	__builtin_memcpy(&u, &a, 8);
	u += (uint64_t)(int64_t)dir;
	__builtin_memcpy(&a, &u, 8);
	return a;
}

constexpr double kFmScale = 16384.0 * (1.0 / 0x1.921fb54442d18p+1);  // the reference's multiplier

// cr, cj: the discriminator's cross terms (exact integers, generic direction); v_fast: the fast path's scaled angle
// (within 1e-9 of an integer).  Returns the reference's fm_dev value under a correctly rounded atan2; *margin_ulps =
// |theta - m| in units of the spacing of doubles below a_k (below 0.06: the reference's own libm may round either way).
TFREC_HD int fm_dev_resolve(double cr, double cj, double v_fast, const double (*coarse)[4], const double (*fine)[4],
			    double *margin_ulps)
{
	const double y = fabs(cj), x = cr;
	const int sgn = cj < 0.0 ? -1 : 1;
	const int k = (int)rint(fabs(v_fast));
	if (k < 1 || k > 16383) {
		// |v| < 0.5 or >= 16383.5: the truncation is 0 / +-16383 whatever the last bits are (integer cross terms below
		// 2^32 keep a generic direction 4e-10 rad = 2e-6 in v off the axes).  Only reached with a widened flag threshold.
		*margin_ulps = 1e30;
		return (int)v_fast;
	}
	const double kd = (double)k;
	// 1. a_k
	double a = kd / kFmScale;
	for (int it = 0; it < 8; it++) {
		const double p = fm_bits_step(a, -1);
		if (p * kFmScale >= kd)
			a = p;
		else
			break;
	}
	for (int it = 0; it < 8; it++) {
		if (a * kFmScale < kd)
			a = fm_bits_step(a, +1);
		else
			break;
	}
	const double gap = a - fm_bits_step(a, -1);  // exact
	// 2. m = a - gap/2;  E = 16384 * (m - k pi / 16384) = 16384 a - k pi - 8192 gap
	const double h1 = kd * TFREC_PI_1;
	const double l1 = __builtin_fma(kd, TFREC_PI_1, -h1);
	const double E = (((16384.0 * a - h1) - l1) - kd * TFREC_PI_2) - 8192.0 * gap;
	const double eps = E * (1.0 / 16384.0);
	// 3. sin / cos(k pi / 16384), double-double
	const int i = k >> 7, j = k & 127;
	const FmDD sA{ coarse[i][0], coarse[i][1] }, cA{ coarse[i][2], coarse[i][3] };
	const FmDD sB{ fine[j][0], fine[j][1] }, cB{ fine[j][2], fine[j][3] };
	const FmDD S = fm_dd_add(fm_dd_mul(sA, cB), fm_dd_mul(cA, sB));
	FmDD nSS = fm_dd_mul(sA, sB);
	nSS.hi = -nSS.hi;
	nSS.lo = -nSS.lo;
	const FmDD C = fm_dd_add(fm_dd_mul(cA, cB), nSS);
	// D = (y C - x S) - eps (y S + x C)
	const double p1 = y * C.hi, e1 = __builtin_fma(y, C.hi, -p1);
	const double p2 = x * S.hi, e2 = __builtin_fma(x, S.hi, -p2);
	const FmDD d = fm_two_sum(p1, -p2);
	const double r = ((d.lo + (e1 - e2)) + (y * C.lo - x * S.lo)) - eps * (y * S.hi + x * C.hi);
	const double D = d.hi + r;
	const double norm = sqrt(x * x + y * y);
	*margin_ulps = fabs(D) / (norm * gap);
	return sgn * (D > 0.0 ? k : k - 1);
}

}  // namespace tfrec
