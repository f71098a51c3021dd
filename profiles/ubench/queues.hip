// How many HIP streams run concurrently?  N streams, one long single-wave kernel each; wall time / kernel time.
// build: hipcc --offload-arch=gfx950 -O3 queues.hip -o queues ; run with and without GPU_MAX_HW_QUEUES=8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(long long *out, long long ticks)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(10);
	out[0] = t0;
}
int main()
{
	long long *d;
	(void)hipMalloc(&d, 64);
	hipStream_t st[16];
	for (int i = 0; i < 16; i++)
		(void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
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
