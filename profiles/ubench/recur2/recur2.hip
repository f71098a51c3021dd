#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "variants.h"
#define STEPS 1024
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","memory"
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, double a1, double a2, const double *gin)
{
	__shared__ double2 pb[64];
	const int ln = threadIdx.x;
	double acc = 0;
	pb[ln] = make_double2(1e-3 * ln, 1e-4 * ln);
	__syncthreads();
	const unsigned lbase = (unsigned)(size_t)pb;  // LDS address
	const double *sb = gin + blockIdx.x * 128;
	long long t0 = clock64();
	for (int s = 0; s < STEPS; s++) {
		double o0, o1, o2;
		double i0 = acc, i1 = acc * 0.5;
#define BODY(A) asm volatile("v_mov_b32 v10, %[i0l]\nv_mov_b32 v11, %[i0h]\nv_mov_b32 v12, %[i1l]\nv_mov_b32 v13, %[i1h]\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v22, 0\nv_mov_b32 v23, 0\n" A "v_mov_b32 %[o0l], v10\nv_mov_b32 %[o0h], v11\nv_mov_b32 %[o1l], v12\nv_mov_b32 %[o1h], v13\n" \
	: [o0l] "=v"(((int*)&o0)[0]), [o0h] "=v"(((int*)&o0)[1]), [o1l] "=v"(((int*)&o1)[0]), [o1h] "=v"(((int*)&o1)[1]) \
	: [i0l] "v"(((int*)&i0)[0]), [i0h] "v"(((int*)&i0)[1]), [i1l] "v"(((int*)&i1)[0]), [i1h] "v"(((int*)&i1)[1]), [a1] "s"(a1), [a2] "s"(a2), [lbase] "v"(lbase), [sbase] "s"(sb) : CLOB)
		if (MODE == 0) BODY(ASM_BARE);
		if (MODE == 1) BODY(ASM_NARROW);
		if (MODE == 2) BODY(ASM_LDS128);
		if (MODE == 3) BODY(ASM_LDS128_NARROW);
		if (MODE == 4) BODY(ASM_LDS64X2);
		if (MODE == 5) BODY(ASM_LDSREAD2);
		if (MODE == 6) BODY("s_load_dwordx16 s[40:55], %[sbase], 0x0\n" ASM_SLOAD);
		if (MODE == 7) BODY("s_load_dwordx16 s[40:55], %[sbase], 0x0\n" ASM_SLOAD_NARROW);
		acc += o0 * 1e-9 + o1 * 1e-12;
	}
	long long t1 = clock64();
	out[2 + threadIdx.x + blockIdx.x * 64] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv)
{
	int only = argc > 1 ? atoi(argv[1]) : -1;
	setvbuf(stdout, 0, _IONBF, 0);
	double *d, *g; long long *c; long long h[1024];
	hipMalloc(&d, (2 + 4096 * 64) * 8); hipMalloc(&c, 4096 * 8); hipMalloc(&g, 4096 * 128 * 8 + 4096);
	hipMemset(d, 0, (2 + 4096 * 64) * 8); hipMemset(g, 0, 4096 * 128 * 8 + 4096);
	const char *names[] = { "bare chain", "bare + exec narrowing", "ds_read_b128 in", "ds_read_b128 in + narrowing", "ds_read_b64 x2 in", "ds_read2_b64 in", "s_load_dwordx16 in", "s_load in + narrowing" };
	for (int blocks : { 1, 1024 }) {
		printf("blocks=%d (one-wave workgroups)\n", blocks);
#define RUN(M) if (only < 0 || only == M) { k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, g); hipDeviceSynchronize(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, g); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, c, 8 * (blocks > 1024 ? 1024 : blocks), hipMemcpyDeviceToHost); \
	printf("  %-32s %.1f ticks per sample, %.2f ns per sample (kernel %.3f ms)\n", names[M], (double)h[0] / STEPS / 64, ms * 1e6 / STEPS / 64, ms); }
		RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
	}
	return 0;
}
