// profiles/ubench/verify_chain.hip -- what one sample of whb_verify_kernel's lane-per-stream recurrence costs a lone wave,
// by parts (round 3).  Every variant runs REPS groups of 8 samples in one wave per workgroup; cycles per sample from
// s_memtime.  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o verify_chain verify_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

template <int V>
__device__ __forceinline__ void group8(const int32_t *d, double bh, double a1, double a2, double &y1, double &y2, double &t1,
				       double &t2, uint32_t &bits)
{
	double t0[8], pp[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		t0[i] = V == 1 ? t1 : bh * (double)d[i];
	pp[0] = V == 1 ? t1 : __builtin_fma(2.0, t1, t0[0]);
#pragma unroll
	for (int i = 1; i < 8; i++)
		pp[i] = V == 1 ? t1 : __builtin_fma(2.0, t0[i - 1], t0[i]);
	double ya = y1, yb = y2;
	unsigned long long cm = 0, co;
	int iyv[8];
	unsigned long long cmv[8];
	double yv[8];
	if (V == 14 || V == 15)
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const double tt2 = i == 0 ? t2 : (i == 1 ? t1 : t0[i - 2]);
		const double m1 = a1 * ya;
		const double m2 = a2 * yb;
		const double s1 = tt2 + m1;
		const double s2 = s1 + pp[i];
		const double y = s2 + m2;
		yv[i] = y;
		if (V == 0) {  // compare into an SGPR pair, carry into the shift
			const int iy = (int)y;
			asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(cm) : "v"(d[i]), "v"(iy));
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		} else if (V == 3) {  // only the conversion
			bits += (uint32_t)(int)y;
		} else if (V == 4) {  // plain C
			bits |= (uint32_t)(d[i] < (int)y) << (i + 8 * (bits & 1));
		} else if (V == 5) {  // vcc, adjacent
			const int iy = (int)y;
			asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(d[i]), "v"(iy) : "vcc");
		} else if (V == 7) {  // d < (int)y  <=>  d >= 0 ? y >= d + 1 : y > d   (two f64 compares, lane masks combined on the SALU)
			const double dd = (double)d[i], dp1 = dd + 1.0;
			unsigned long long c1, c2, ng;
			asm("v_cmp_ge_f64_e64 %0, %1, %2" : "=s"(c1) : "v"(y), "v"(dp1));
			asm("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(c2) : "v"(y), "v"(dd));
			asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(ng) : "v"(d[i]));
			cm = c1 | (c2 & ng);
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		} else if (V == 8) {  // one f64 compare against a threshold the pre-pass made: d >= 0 ? d + 1 : nextup(d)
			const double dd = (double)d[i];
			const double dp1 = dd + 1.0;
			const unsigned long long nb = (unsigned long long)__double_as_longlong(dd) - 1ull;  // nextup of a negative double
			const double th = d[i] >= 0 ? dp1 : __longlong_as_double((long long)nb);
			asm("v_cmp_ge_f64_e64 %0, %1, %2" : "=s"(cm) : "v"(y), "v"(th));
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		} else if (V == 9) {  // only one f64 compare + addc (what a free threshold would cost)
			asm("v_cmp_ge_f64_e64 %0, %1, %2" : "=s"(cm) : "v"(y), "v"(t0[i]));
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		} else if (V == 10) {  // only v_trunc_f64
			bits += (uint32_t)__double2loint(__builtin_trunc(y));
		} else if (V == 11 || V == 12) {  // convert now, compare + shift at the end of the group
			iyv[i] = (int)y;
		} else if (V == 13) {  // f64 compare now (threshold free), shift at the end of the group
			asm("v_cmp_ge_f64_e64 %0, %1, %2" : "=s"(cmv[i]) : "v"(y), "v"(t0[i]));
		} else if (V == 6) {  // compare in the double domain: d < (int)y  <=>  (double)d < trunc(y)
			const double ty = __builtin_trunc(y);
			asm("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(cm) : "v"((double)d[i]), "v"(ty));
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		}
		yb = ya;
		ya = y;
	}
	if (V == 14 || V == 15) {  // the eight conversions TOGETHER behind the chain (a scheduling barrier keeps them there)
		__builtin_amdgcn_sched_barrier(0);
		int iy[8];
#pragma unroll
		for (int i = 0; i < 8; i++)
			iy[i] = (int)yv[i];
		__builtin_amdgcn_sched_barrier(0);
		if (V == 14) {
#pragma unroll
			for (int i = 0; i < 8; i++)
				bits += (uint32_t)iy[i];
		} else {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(cmv[i]) : "v"(d[i]), "v"(iy[i]));
			}
#pragma unroll
			for (int i = 0; i < 8; i++)
				asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cmv[i]));
		}
		__builtin_amdgcn_sched_barrier(0);
	}
	if (V == 11) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(cm) : "v"(d[i]), "v"(iyv[i]));
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cm));
		}
	}
	if (V == 12) {  // all eight compares into ONE mask register by lane-wise packing: no SGPR round trip per sample
		uint32_t w = 0;
#pragma unroll
		for (int i = 0; i < 8; i++)
			w |= (uint32_t)(d[i] < iyv[i]) << (7 - i);
		bits = (bits << 8) | w;
	}
	if (V == 13) {
#pragma unroll
		for (int i = 0; i < 8; i++)
			asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(co) : "s"(cmv[i]));
	}
	y1 = ya;
	y2 = yb;
	t2 = t0[6];
	t1 = t0[7];
}

template <int V, int NS>
__global__ __launch_bounds__(64) void k(const int32_t *in, int reps, double a1, double a2, double bh, unsigned long long *out, double *sink)
{
	int32_t A[NS][8];
	double y1[NS], y2[NS], t1[NS], t2[NS];
	uint32_t bits[NS];
#pragma unroll
	for (int s = 0; s < NS; s++) {
#pragma unroll
		for (int i = 0; i < 8; i++)
			A[s][i] = in[(threadIdx.x * NS + s) * 8 + i];
		y1[s] = 1e8 + threadIdx.x;
		y2[s] = 1e8;
		t1[s] = 3.0;
		t2[s] = 2.0;
		bits[s] = 0;
	}
	const unsigned long long c0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) {
#pragma unroll
		for (int s = 0; s < NS; s++)
			group8<V>(A[s], bh, a1, a2, y1[s], y2[s], t1[s], t2[s], bits[s]);
#pragma unroll
		for (int s = 0; s < NS; s++)
			A[s][0] += (int)(bits[s] & 1u) + (r & 1);  // keeps the inputs loop-variant (a static index: a dynamic one is a 35-instruction select cascade)
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	double acc = 0;
#pragma unroll
	for (int s = 0; s < NS; s++)
		acc += y1[s] + y2[s] + bits[s];
	if (acc == 1.2345)
		*sink = acc;
	if (threadIdx.x == 0)
		out[blockIdx.x] = c1 - c0;
}

template <int V, int NS>
static void run(const char *name)
{
	const int reps = 4000, blocks = 64;
	int32_t *d_in;
	unsigned long long *d_out, h[64];
	double *sink;
	hipMalloc(&d_in, 64 * 8 * 8 * 4);
	int32_t hin[64 * 8 * 8];
	for (int i = 0; i < 64 * 8 * 8; i++)
		hin[i] = (i * 2654435761u) >> 4;
	hipMemcpy(d_in, hin, sizeof(hin), hipMemcpyHostToDevice);
	hipMalloc(&d_out, sizeof(h));
	hipMalloc(&sink, 8);
	hipLaunchKernelGGL((k<V, NS>), dim3(blocks), dim3(64), 0, 0, d_in, reps, 1.9997, -0.99973, 7.5e-9, d_out, sink);
	if (hipDeviceSynchronize() != hipSuccess) { printf("%s: failed\n", name); return; }
	hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
	unsigned long long m = h[0];
	for (int i = 1; i < blocks; i++) m = h[i] < m ? h[i] : m;
	printf("{\"variant\": \"%s\", \"streams_per_lane\": %d, \"cycles_per_sample_per_stream\": %.2f, \"cycles_per_group_iteration\": %.1f}\n", name, NS,
	       (double)m / reps / 8 / NS, (double)m / reps);
}

int main()
{
	setvbuf(stdout, NULL, _IONBF, 0);
	run<1, 1>("chain only (5 fp64 ops)");
	run<2, 1>("chain + feed-forward (8 ops)");
	run<3, 1>("+ cvt_i32_f64");
	run<0, 1>("+ cmp -> sgpr pair -> addc (kernel)");
	run<5, 1>("+ cmp vcc, addc vcc adjacent");
	run<4, 1>("+ compare in plain C");
	run<6, 1>("+ trunc, cmp_lt_f64 -> addc");
	run<7, 1>("+ two f64 compares + salu and/or -> addc");
	run<8, 1>("+ threshold by pre-pass, one f64 compare -> addc");
	run<9, 1>("+ one f64 compare -> addc only");
	run<10, 1>("+ v_trunc_f64 only");
	run<14, 1>("chain of 8, then the 8 cvt_i32_f64 together (sched_barrier)");
	run<15, 1>("chain of 8, then 8 cvt, 8 cmp, 8 addc, each kind together");
	run<11, 1>("cvt per sample, 8 x (cmp -> sgpr -> addc) at the group's end");
	run<12, 1>("cvt per sample, 8 compares packed in plain C at the group's end");
	run<13, 1>("f64 compare per sample into 8 sgpr pairs, 8 addc at the group's end");
	run<11, 2>("cvt / deferred cmp, 2 streams per lane");
	run<12, 2>("cvt / packed C, 2 streams per lane");
	run<7, 2>("two-compare form, 2 streams per lane");
	run<8, 2>("threshold form, 2 streams per lane");
	run<8, 3>("threshold form, 3 streams per lane");
	run<8, 4>("threshold form, 4 streams per lane");
	run<0, 2>("kernel form, 2 streams per lane");
	run<0, 3>("kernel form, 3 streams per lane");
	run<6, 2>("trunc form, 2 streams per lane");
	run<1, 2>("chain only, 2 streams per lane");
	return 0;
}
