// profiles/ubench/ipollute.hip -- an instruction-cache polluter to run BESIDE bench.py (another process): 256 one-wave
// workgroups, each walking a loop body of <KB> kilobytes of dependent v_fma_f32 (one instruction every ~8 cycles per wave:
// a few per cent of one SIMD's issue slots per CU) for <seconds>.  `ipollute 2 20` is the control for `ipollute 48 20`: the
// same instructions at the same rate, 2 KB of code instead of 48.   hipcc --offload-arch=gfx950 -O3 ipollute.hip -o ipollute
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

template <int UNROLL>
__global__ __launch_bounds__(64) void body(float *out, int iters, float c, float d)
{
	float x = (float)threadIdx.x;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < UNROLL; u++)
			asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
	}
	out[blockIdx.x * 64 + threadIdx.x] = x;
}

int main(int argc, char **argv)
{
	const int kb = argc > 1 ? atoi(argv[1]) : 48;
	const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
	const int grid = argc > 3 ? atoi(argv[3]) : 256;
	float *out;
	(void)hipMalloc(&out, 1 << 20);
	const int total = 1 << 18;  // instructions per wave and launch (~1 ms)
	const auto t0 = std::chrono::steady_clock::now();
	long launches = 0;
	while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
		for (int r = 0; r < 8; r++) {
			if (kb >= 48)
				hipLaunchKernelGGL(body<6144>, dim3(grid), dim3(64), 0, 0, out, total / 6144, 1.0001f, 0.5f);
			else if (kb >= 24)
				hipLaunchKernelGGL(body<3072>, dim3(grid), dim3(64), 0, 0, out, total / 3072, 1.0001f, 0.5f);
			else
				hipLaunchKernelGGL(body<256>, dim3(grid), dim3(64), 0, 0, out, total / 256, 1.0001f, 0.5f);
			launches++;
		}
		(void)hipDeviceSynchronize();
	}
	printf("ipollute %d KB grid %d: %ld launches in %.1f s\n", kb, grid, launches, seconds);
	return 0;
}
