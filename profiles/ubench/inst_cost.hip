// profiles/ubench/inst_cost.hip -- issue cost of single instructions for a LONE wave and for 4 waves per SIMD (round 3: which
// conversions / compares are slow).  Each variant: REPS x 32 independent instances of one instruction, s_memtime around.
// build: hipcc --offload-arch=gfx950 -O2 -o inst_cost inst_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R32(X) R8(X) R8(X) R8(X) R8(X)

template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int reps, double seed)
{
	double d[8], e[8];
	float f[8];
	int n[8];
	unsigned long long q[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		d[i] = seed * (threadIdx.x + 1 + i) + 0.37;
		e[i] = d[i] * 1.5;
		f[i] = (float)d[i];
		n[i] = threadIdx.x * 77 + i;
		q[i] = (unsigned long long)n[i] * 1234567ull;
	}
	unsigned long long cm = 0;
	int sl = 0;
	const unsigned long long c0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) {
		if (V == 0) {
#define X(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(d[i]));
			R32(X)
#undef X
		} else if (V == 1) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
			R32(X)
#undef X
		} else if (V == 2) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(f[i]));
			R32(X)
#undef X
		} else if (V == 3) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(n[i]));
			R32(X)
#undef X
		} else if (V == 4) {
#define X(i) asm volatile("v_trunc_f64 %0, %1" : "=v"(e[i]) : "v"(d[i]));
			R32(X)
#undef X
		} else if (V == 5) {
#define X(i) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(cm) : "v"(d[i]), "v"(e[i]));
			R32(X)
#undef X
		} else if (V == 6) {
#define X(i) asm volatile("v_min_f64 %0, %1, %2" : "=v"(e[i]) : "v"(d[i]), "v"(e[i]));
			R32(X)
#undef X
		} else if (V == 7) {
#define X(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(e[i]) : "v"(d[i]));
			R32(X)
#undef X
		} else if (V == 8) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) & 7]));
			R32(X)
#undef X
		} else if (V == 9) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(q[i]) : "v"(q[i]));
			R32(X)
#undef X
		} else if (V == 10) {
#define X(i) asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(cm) : "v"(q[i]), "v"(q[(i + 1) & 7]));
			R32(X)
#undef X
		} else if (V == 11) {
#define X(i) asm volatile("v_cmp_lt_i32 %0, %1, %2" : "=s"(cm) : "v"(n[i]), "v"(n[(i + 1) & 7]));
			R32(X)
#undef X
		} else if (V == 12) {
#define X(i) asm volatile("v_rndne_f64 %0, %1" : "=v"(e[i]) : "v"(d[i]));
			R32(X)
#undef X
		} else if (V == 13) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
			R32(X)
#undef X
		} else if (V == 14) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(n[i]) : "v"(n[i]), "v"(n[(i + 1) & 7]), "s"(cm));
			R32(X)
#undef X
		} else if (V == 15) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(n[i]) : "v"(n[(i + 1) & 7]));
			R32(X)
#undef X
		} else if (V == 16) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sl) : "v"(n[i]));
			R32(X)
#undef X
		} else if (V == 17) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[i]));
			R32(X)
#undef X
		}
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	double acc = (double)cm + sl;
#pragma unroll
	for (int i = 0; i < 8; i++)
		acc += d[i] + e[i] + f[i] + n[i] + (double)q[i];
	if (acc == 1.2345)
		out[1000] = 1;
	if ((threadIdx.x & 63) == 0)
		out[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int V>
static void run(const char *name)
{
	const int reps = 2000;
	unsigned long long *d_out, h[8192];
	hipMalloc(&d_out, sizeof(h));
	for (int W : { 1, 4 }) {
		const int blocks = 256 * W;  // one block of 4 waves per CU and W
		hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, reps, 1.001);
		if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
		hipMemcpy(h, d_out, blocks * 4 * 8, hipMemcpyDeviceToHost);
		unsigned long long m = h[0];
		for (int i = 1; i < blocks * 4; i++) m = h[i] < m ? h[i] : m;
		printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.2f, \"cycles_per_inst_per_simd\": %.2f}\n", name, W,
		       (double)m / reps / 32, (double)m / reps / 32 / W);
	}
	hipFree(d_out);
}

int main()
{
	setvbuf(stdout, NULL, _IONBF, 0);
	run<0>("v_cvt_i32_f64");
	run<1>("v_cvt_f32_f64");
	run<2>("v_cvt_i32_f32");
	run<3>("v_cvt_f64_i32");
	run<13>("v_cvt_f64_f32");
	run<4>("v_trunc_f64");
	run<12>("v_rndne_f64");
	run<5>("v_cmp_lt_f64");
	run<6>("v_min_f64");
	run<7>("v_rcp_f64");
	run<8>("v_mul_hi_u32");
	run<9>("v_lshlrev_b64");
	run<10>("v_cmp_lt_u64");
	run<11>("v_cmp_lt_i32");
	run<14>("v_cndmask_b32");
	run<15>("v_mov_b32_dpp row_shr");
	run<16>("v_readlane_b32");
	run<17>("ds_bpermute_b32 + wait");
	return 0;
}
