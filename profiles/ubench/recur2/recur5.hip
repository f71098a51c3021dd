#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "variants4.h"
#define STEPS 1024
#define CLOB "memory","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, double a1, double a2, int *bad)
{
	__shared__ double2 pb[64];
	const int ln = threadIdx.x;
	double y1 = 0.25, y2 = 0.125, acc = 0;
	long long t0 = clock64();
	int nbad = 0;
	for (int s = 0; s < STEPS; s++) {
		pb[ln] = make_double2(1e-3 * ln + s * 0.37 + acc * 1e-7, 1e-4 * ln - s * 0.11);
		__syncthreads();
		double2 in[4];
#pragma unroll
		for (int j = 0; j < 4; j++)
			in[j] = pb[16 * j + (ln & 15)];
		double z[4], o1, o2;
#define CHAIN(A) asm volatile( \
	"v_mov_b64 v[16:17], %[y1]\nv_mov_b64 v[14:15], %[y2]\nv_mov_b64 v[22:23], 1.0\n" \
	"v_mov_b64 v[24:25], %[p0]\nv_mov_b64 v[26:27], %[b0]\nv_mov_b64 v[28:29], %[p1]\nv_mov_b64 v[30:31], %[b1]\n" \
	"v_mov_b64 v[32:33], %[p2]\nv_mov_b64 v[34:35], %[b2]\nv_mov_b64 v[36:37], %[p3]\nv_mov_b64 v[38:39], %[b3]\n" \
	"s_nop 1\n" A \
	"v_mov_b64 %[z0], v[40:41]\nv_mov_b64 %[z1], v[42:43]\nv_mov_b64 %[z2], v[44:45]\nv_mov_b64 %[z3], v[46:47]\nv_mov_b64 %[o1], v[16:17]\nv_mov_b64 %[o2], v[14:15]\n" \
	: [z0] "=v"(z[0]), [z1] "=v"(z[1]), [z2] "=v"(z[2]), [z3] "=v"(z[3]), [o1] "=v"(o1), [o2] "=v"(o2) \
	: [y1] "v"(y1), [y2] "v"(y2), [a1] "s"(a1), [a2] "s"(a2), [p0] "v"(in[0].x), [b0] "v"(in[0].y), [p1] "v"(in[1].x), [b1] "v"(in[1].y), \
	  [p2] "v"(in[2].x), [b2] "v"(in[2].y), [p3] "v"(in[3].x), [b3] "v"(in[3].y) : CLOB)
		if (MODE == 0) CHAIN(ASM_DPP);
		if (MODE == 1) CHAIN(ASM_DPP_NOCAP);
		const int m = ln & 3;
		const double ym = m == 0 ? z[0] : m == 1 ? z[1] : m == 2 ? z[2] : z[3];
		if (s < 64) {  // check against the plain chain
			double c1 = y1, c2 = y2, mine = 0;
			for (int q = 0; q < 64; q++) {
				const double2 v = pb[q];
				const double y = ((v.y + a1 * c1) + v.x) + a2 * c2;
				c2 = c1; c1 = y;
				if (q == ln) mine = y;
			}
			if (MODE == 0 && mine != ym) nbad++;
			if (c1 != o1 || c2 != o2) nbad += 1000;
		}
		y1 = o1; y2 = o2;
		acc += (MODE == 0 ? ym : 0.0) + o1;
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
	const char *names[] = { "DPP inputs + masked-DPP capture", "DPP inputs, no capture" };
	for (int blocks : { 1, 1024 }) {
		printf("blocks=%d (one-wave workgroups)\n", blocks);
#define RUN(M) if (only < 0 || only == M) { hipMemset(bad, 0, 4); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, bad); hipDeviceSynchronize(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025, bad); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, c, 8 * (blocks > 1024 ? 1024 : blocks), hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); \
	printf("  %-32s %.1f ticks per sample (incl. step overhead), %.2f ns per sample, mismatches %d\n", names[M], (double)h[0] / STEPS / 64, ms * 1e6 / STEPS / 64, hb); }
		RUN(0) RUN(1)
	}
	return 0;
}
