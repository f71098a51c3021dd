// How many HIP streams run concurrently?  N streams, one long single-wave kernel each; wall time / kernel time.
// build: hipcc --offload-arch=gfx950 -O3 queues.hip -o queues ; run with and without GPU_MAX_HW_QUEUES=8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
__global__ void spin(long long *out, long long ticks)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(10);
	out[0] = t0;
}
int main(int argc, char **argv)
{
	const int prio_mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: default priority, 1: all high, 2: first low + rest high
	long long *d;
	(void)hipMalloc(&d, 64);
	hipStream_t st[16];
	int lo = 0, hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
	printf("priority mode %d (range %d..%d)\n", prio_mode, lo, hi);
	for (int i = 0; i < 16; i++) {
		if (prio_mode == 0)
			(void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
		else
			(void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (prio_mode == 2 && i == 0) ? lo : hi);
	}
	spin<<<1, 64>>>(d, 1000);
	(void)hipDeviceSynchronize();
	for (int n : { 1, 2, 4, 5, 6, 8, 12, 16 }) {
		const auto t0 = std::chrono::steady_clock::now();
		for (int rep = 0; rep < 3; rep++)  // three dependent kernels per stream
			for (int i = 0; i < n; i++)
				spin<<<1, 64, 0, st[i]>>>(d, 200000);  // 2 ms at 100 MHz
		(void)hipDeviceSynchronize();
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		printf("%2d streams x 3 kernels of 2 ms: %.2f ms wall\n", n, ms);
	}
	return 0;
}
