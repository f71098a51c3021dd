// Micro-benchmark: what does ONE wave per SIMD pay per instruction?  (serial demodulator chains run like this)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lone_wave.hip -o lone_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b)
{
	double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
	float f0 = (float)x0, f1 = f0 + 1;
	int i0 = (int)x0, i1 = i0 + 1;
	long long t0 = clock64();
	for (int n = 0; n < N; n++) {
		if (MODE == 0) { x0 = x0 * a; x0 = x0 + b; x0 = x0 * a; x0 = x0 + b; }            // 4 dependent DP
		if (MODE == 1) { x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; }            // 4 independent DP
		if (MODE == 2) { f0 = f0 * (float)a; f0 = f0 + (float)b; f0 = f0 * (float)a; f0 = f0 + (float)b; }  // dep SP
		if (MODE == 3) { i0 = i0 * 3 + 1; i0 = i0 ^ (i0 >> 3); i0 = i0 + 7; i0 = i0 ^ (i0 << 2); }          // dep int (6 ops)
		if (MODE == 4) { i0 = i0 * 3 + 1; i1 = i1 ^ (i1 >> 3); i0 = i0 + 7; i1 = i1 ^ (i1 << 2); }
		if (MODE == 5) { x0 = ((b * x1 + a * x0) + (b * x2 + a * x3)) + a * x1; x1 = x0 * b; }  // biquad-like
	}
	long long t1 = clock64();
	out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3 + f0 + f1 + i0 + i1;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
	double *d; long long *c; long long h[1024]; (void)0;
	hipMalloc(&d, 1024 * 64 * 8); hipMalloc(&c, 1024 * 8);
	hipMemset(d, 0, 1024 * 64 * 8);
	const char *names[] = { "4 dependent f64 (mul,add,mul,add)", "4 independent f64", "4 dependent f32", "dependent int chain (~7 ops)",
				"2 independent int chains", "biquad-like expr (6 mul, 4 add)" };
	for (int blocks : { 1, 80, 1024, 4096 }) {
		printf("blocks=%d (64-thread blocks)\n", blocks);
#define RUN(M) { k<M><<<blocks, 64>>>(d, c, 1.0000001, 1e-9); hipDeviceSynchronize(); k<M><<<blocks, 64>>>(d, c, 1.0000001, 1e-9); hipMemcpy(h, c, 8 * (blocks > 1024 ? 1024 : blocks), hipMemcpyDeviceToHost); \
	printf("  %-40s %.1f clock64 ticks per loop iteration\n", names[M], (double)h[0] / N); }
		RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
	}
	// clock64 rate
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0); k<0><<<1, 64>>>(d, c, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
	printf("clock64: %lld ticks in %.3f ms kernel -> %.1f MHz (lower bound)\n", h[0], ms, h[0] / ms / 1e3);
	return 0;
}
