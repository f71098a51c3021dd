// profiles/ubench/valu_issue.hip -- measured VALU/SALU issue cost per instruction class on gfx950, with 1..8 waves per
// SIMD: the "cycles per issue" of bench.py's roofline.valu block.
// Every wave runs REPS x 64 independent instructions of one class (8 accumulators, so no dependency stall with >= 1
// wave); a workgroup is 256 threads = one wave per SIMD of its CU, W workgroups per CU -> W waves per SIMD.
// cycles per instruction per SIMD = (max end - min start of s_memtime over the waves of a SIMD-equivalent) / (REPS*64*W).
// build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int CLS>
__global__ __launch_bounds__(256) void issue_kernel(unsigned long long *t, int reps, float *sink)
{
	float a[8];
	double d[8];
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p[8];
	int n[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		a[i] = threadIdx.x * 1e-3f + i;
		d[i] = threadIdx.x * 1e-3 + i;
		p[i] = f2{ a[i], a[i] + 1 };
		n[i] = threadIdx.x + i;
	}
	const float c = 1.0001f;
	const double cd = 1.0001;
	const f2 cp = { 1.0001f, 0.9999f };
	int s0 = reps, s1 = 3, s2 = 5, s3 = 7;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++) {
		if (CLS == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
			REP64(X)
#undef X
		} else if (CLS == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
			REP64(X)
#undef X
		} else if (CLS == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
			REP64(X)
#undef X
		} else if (CLS == 3) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
			REP64(X)
#undef X
		} else if (CLS == 4) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
			REP64(X)
#undef X
		} else if (CLS == 5) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
			REP64(X)
#undef X
		} else if (CLS == 6) {  // dependent fp64 chain: latency of one wave's back-to-back dependent v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(cd));
			REP64(X)
#undef X
		} else if (CLS == 7) {  // SALU
#define X(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
			REP64(X)
#undef X
		} else if (CLS == 8) {  // v_cvt_f64_i32 (conversions: quarter rate?)
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(n[i]));
			REP64(X)
#undef X
		} else if (CLS == 9) {  // v_mul_lo_u32 (integer multiply)
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
			REP64(X)
#undef X
		} else if (CLS == 10) {  // dependent fp32 chain
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(c));
			REP64(X)
#undef X
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float acc = 0;
#pragma unroll
	for (int i = 0; i < 8; i++)
		acc += a[i] + (float)d[i] + p[i].x + p[i].y + n[i];
	acc += s0 + s2 + s3;
	if (acc == 12345.678f)
		*sink = acc;
	if ((threadIdx.x & 63) == 0) {
		const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
		t[2 * w] = t0;
		t[2 * w + 1] = t1;
	}
}

template <int CLS>
static void run(const char *name, int W, int n_cu)
{
	const int reps = 2000;
	const int blocks = n_cu * W;
	unsigned long long *d_t;
	float *sink;
	if (hipMalloc(&d_t, blocks * 4 * 2 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(issue_kernel<CLS>, dim3(blocks), dim3(256), 0, 0, d_t, 10, sink);  // warm-up
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(issue_kernel<CLS>, dim3(blocks), dim3(256), 0, 0, d_t, reps, sink);
	hipEventRecord(e1);
	{ hipError_t e_ = hipDeviceSynchronize(); if (e_ != hipSuccess) { fprintf(stderr, "%s W=%d: %s\n", name, W, hipGetErrorString(e_)); exit(1); } }
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> t(blocks * 8);
	hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
	// per-wave cycles (s_memtime = shader clock): median over waves
	std::vector<double> per;
	for (int w = 0; w < blocks * 4; w++)
		per.push_back((double)(t[2 * w + 1] - t[2 * w]));
	std::sort(per.begin(), per.end());
	const double med = per[per.size() / 2];
	const double n_inst = (double)reps * 64;
	// a SIMD hosts W waves: it issued W * n_inst instructions in `med` cycles (all waves run concurrently if they fit)
	printf("{\"class\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.3f, \"cycles_per_inst_per_simd\": %.3f, "
	       "\"kernel_ms\": %.4f, \"clock_ghz\": %.3f}\n",
	       name, W, med / n_inst, med / n_inst / W, ms, med / (ms * 1e6));
	hipFree(d_t);
	hipFree(sink);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main()
{
	setvbuf(stdout, NULL, _IONBF, 0);
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int n_cu = p.multiProcessorCount;
	fprintf(stderr, "%s, %d CUs\n", p.name, n_cu);
	for (int W : { 1, 2, 4, 8 }) {
		run<0>("v_fma_f32", W, n_cu);
		run<1>("v_pk_fma_f32", W, n_cu);
		run<2>("v_fma_f64", W, n_cu);
		run<3>("v_add_u32", W, n_cu);
		run<4>("v_add_f64", W, n_cu);
		run<5>("v_mul_f64", W, n_cu);
		run<8>("v_cvt_f64_i32", W, n_cu);
		run<9>("v_mul_lo_u32", W, n_cu);
		run<7>("s_add_u32", W, n_cu);
	}
	run<6>("v_fma_f64 dependent chain", 1, n_cu);
	run<10>("v_fma_f32 dependent chain", 1, n_cu);
	return 0;
}
