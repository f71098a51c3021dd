#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "variants2.h"
#define STEPS 1024
#define CLOB "scc","memory","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87"
// chain: in y1 (y(-1)), y2 (y(-2)); out: the four rotating pairs
#define CHAIN(A) asm volatile("v_mov_b32 v16, %[i0l]\nv_mov_b32 v17, %[i0h]\nv_mov_b32 v14, %[i1l]\nv_mov_b32 v15, %[i1h]\n" A \
	"v_mov_b32 %[p0l], v10\nv_mov_b32 %[p0h], v11\nv_mov_b32 %[p1l], v12\nv_mov_b32 %[p1h], v13\nv_mov_b32 %[p2l], v14\nv_mov_b32 %[p2h], v15\nv_mov_b32 %[p3l], v16\nv_mov_b32 %[p3h], v17\n" \
	: [p0l] "=v"(pl[0]), [p0h] "=v"(ph[0]), [p1l] "=v"(pl[1]), [p1h] "=v"(ph[1]), [p2l] "=v"(pl[2]), [p2h] "=v"(ph[2]), [p3l] "=v"(pl[3]), [p3h] "=v"(ph[3]) \
	: [i0l] "v"(__double2loint(y1)), [i0h] "v"(__double2hiint(y1)), [i1l] "v"(__double2loint(y2)), [i1h] "v"(__double2hiint(y2)), [a1] "s"(a1), [a2] "s"(a2), [lbase] "v"(lbase) : CLOB)
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, double a1, double a2, int *bad)
{
	__shared__ double2 pb[64];
	__shared__ double yl[64];
	const int ln = threadIdx.x;
	double y1 = 0.25, y2 = 0.125, acc = 0;
	const unsigned lbase = (unsigned)(size_t)pb;
	long long t0 = clock64();
	int nbad = 0;
	for (int s = 0; s < STEPS; s++) {
		pb[ln] = make_double2(1e-3 * ln + s * 0.37 + acc * 1e-7, 1e-4 * ln - s * 0.11);
		__syncthreads();
		int pl[4], ph[4];
		if (MODE == 0) CHAIN(ASM_N12_S0);
		if (MODE == 1) CHAIN(ASM_N12_S1);
		if (MODE == 2) CHAIN(ASM_N12_S2);
		if (MODE == 3) CHAIN(ASM_N15_S0);
		if (MODE == 4) CHAIN(ASM_W12_S0);
		if (MODE == 5) CHAIN(ASM_N12_S0_G8);
		if (MODE == 6) CHAIN(ASM_N8_S0_PER);
		double ym;
		if (MODE == 5) {
			ym = 0;  // (group of 8 with 4 pairs cannot keep all: timing only)
		} else {
			const int r = ln & 3;
			const int lo = r == 0 ? pl[0] : r == 1 ? pl[1] : r == 2 ? pl[2] : pl[3];
			const int hi = r == 0 ? ph[0] : r == 1 ? ph[1] : r == 2 ? ph[2] : ph[3];
			ym = __hiloint2double(hi, lo);
		}
		if (MODE != 4 && MODE != 5 && s < 64) {  // check against the plain chain
			double c1 = y1, c2 = y2, mine = 0;
			for (int q = 0; q < 64; q++) {
				const double2 v = pb[q];
				const double y = ((v.y + a1 * c1) + v.x) + a2 * c2;
				c2 = c1; c1 = y;
				if (q == ln) mine = y;
			}
			if (mine != ym) nbad++;
		}
		yl[ln] = ym;
		__syncthreads();
		if (MODE == 4 || MODE == 5) { y1 = __hiloint2double(ph[3], pl[3]); y2 = __hiloint2double(ph[2], pl[2]); }
		else { y1 = yl[63]; y2 = yl[62]; }
		acc += ym;
		__syncthreads();
	}
	long long t1 = clock64();
	out[2 + threadIdx.x + blockIdx.x * 64] = acc;
	if (nbad) atomicAdd(bad, nbad);
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv)
{
	int only = argc > 1 ? atoi(argv[1]) : -1;
	setvbuf(stdout, 0, _IONBF, 0);
	double *d; long long *c; long long h[1024]; int *bad, hb;
	hipMalloc(&d, (2 + 4096 * 64) * 8); hipMalloc(&c, 4096 * 8); hipMalloc(&bad, 4);
	hipMemset(d, 0, (2 + 4096 * 64) * 8);
	const char *names[] = { "narrow g4 D12 sched0", "narrow g4 D12 sched1", "narrow g4 D12 sched2", "narrow g4 D15 sched0", "no narrowing D12 sched0", "narrow g8 D12 (timing only)", "narrow g4 D8 wait per sample" };
	for (int blocks : { 1, 1024 }) {
		printf("blocks=%d (one-wave workgroups)\n", blocks);
#define RUN(M) if (only < 0 || only == M) { hipMemset(bad, 0, 4); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, bad); hipDeviceSynchronize(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, bad); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, c, 8 * (blocks > 1024 ? 1024 : blocks), hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); \
	printf("  %-32s %.1f ticks per sample (incl. step overhead), %.2f ns per sample, mismatches %d\n", names[M], (double)h[0] / STEPS / 64, ms * 1e6 / STEPS / 64, hb); }
		RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
	}
	return 0;
}
