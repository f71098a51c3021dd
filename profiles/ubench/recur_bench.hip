// Micro-benchmark: the WHB decision-level recurrence (whb_demod_kernel step 2) as ONE wave runs it.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off recur_bench.hip -o recur_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define STEPS 2048
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, double a1, double a2)
{
	__shared__ double2 pb[64];
	__shared__ double yl[64];
	const int ln = threadIdx.x;
	double y1 = out[0], y2 = out[1];
	double acc = 0;
	long long t0 = clock64();
	for (int s = 0; s < STEPS; s++) {
		pb[ln] = make_double2(1e-3 * ln + s, 1e-4 * ln);
		__syncthreads();
		if (MODE == 0) {
#pragma unroll 4
			for (int k = 0; k < 64; k++) {
				const double2 v = pb[k];
				const double y = ((v.y + a1 * y1) + v.x) + a2 * y2;
				yl[k] = y;
				y2 = y1;
				y1 = y;
			}
		} else if (MODE == 1) {  // prefetch 8
			double2 v[8], w[8];
#pragma unroll
			for (int q = 0; q < 8; q++) v[q] = pb[q];
			for (int k = 0; k < 64; k += 8) {
#pragma unroll
				for (int q = 0; q < 8; q++) w[q] = pb[(k + 8 + q) & 63];
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const double y = ((v[q].y + a1 * y1) + v[q].x) + a2 * y2;
					yl[k + q] = y;
					y2 = y1;
					y1 = y;
				}
#pragma unroll
				for (int q = 0; q < 8; q++) v[q] = w[q];
			}
		} else if (MODE == 2) {  // no LDS: readlane
			const double2 mine = pb[ln];
			double ym = 0;
#pragma unroll 8
			for (int k = 0; k < 64; k++) {
				double vx, vy;
				{
					int lo = __builtin_amdgcn_readlane((int)__double2loint(mine.x), k), hi = __builtin_amdgcn_readlane(__double2hiint(mine.x), k);
					vx = __hiloint2double(hi, lo);
					lo = __builtin_amdgcn_readlane((int)__double2loint(mine.y), k), hi = __builtin_amdgcn_readlane(__double2hiint(mine.y), k);
					vy = __hiloint2double(hi, lo);
				}
				const double y = ((vy + a1 * y1) + vx) + a2 * y2;
				ym = ln == k ? y : ym;
				y2 = y1;
				y1 = y;
			}
			yl[ln] = ym;
		} else if (MODE == 3) {  // the bare dependent chain (no inputs, no outputs)
#pragma unroll 8
			for (int k = 0; k < 64; k++) {
				const double y = ((1e-4 + a1 * y1) + 1e-3) + a2 * y2;
				y2 = y1;
				y1 = y;
			}
		} else if (MODE == 4) {  // fully unrolled, all reads first
			double2 v[64];
#pragma unroll
			for (int q = 0; q < 64; q++) v[q] = pb[q];
#pragma unroll
			for (int q = 0; q < 64; q++) {
				const double y = ((v[q].y + a1 * y1) + v[q].x) + a2 * y2;
				yl[q] = y;
				y2 = y1;
				y1 = y;
			}
		}
		__syncthreads();
		acc += yl[ln];
	}
	long long t1 = clock64();
	out[2 + threadIdx.x + blockIdx.x * 64] = acc + y1;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
	double *d; long long *c; long long h[1024];
	hipMalloc(&d, (2 + 4096 * 64) * 8); hipMalloc(&c, 4096 * 8);
	hipMemset(d, 0, (2 + 4096 * 64) * 8);
	const char *names[] = { "as shipped (unroll 4, LDS read per sample)", "prefetch 8", "readlane inputs, cndmask outputs", "bare chain", "64 reads first, unrolled" };
	for (int blocks : { 1, 1024, 4096 }) {
		printf("blocks=%d (one-wave workgroups)\n", blocks);
#define RUN(M) { k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025); hipDeviceSynchronize(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); k<M><<<blocks, 64>>>(d, c, 1.9, -0.9025); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, c, 8 * (blocks > 1024 ? 1024 : blocks), hipMemcpyDeviceToHost); \
	printf("  %-45s %.1f ticks per sample, %.2f ns per sample (kernel %.3f ms)\n", names[M], (double)h[0] / STEPS / 64, ms * 1e6 / STEPS / 64, ms); }
		RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
	}
	return 0;
}
