// profiles/ubench/icache.hip -- does the instruction cache bind when several kernels with long unrolled loops share a CU?
// Kernel<ID, UNROLL>: one wave per SIMD (256 x 256 threads), a loop whose body is UNROLL x 8 v_fma_f32 (8 bytes each) on
// CHAINS independent accumulators.  Four copies run on four streams, either the SAME function four times (one copy of the
// code in the cache) or four DIFFERENT instantiations (four copies): the instruction mix and the issue contention are the
// same, only the code footprint differs.   hipcc --offload-arch=gfx950 -O3 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ID, int UNROLL, int CHAINS>
__global__ __launch_bounds__(256) void kern(float *out, int iters, float c, float d)
{
	float a[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		a[i] = (float)(threadIdx.x + i + ID);
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				float &x = a[CHAINS == 8 ? i : i % CHAINS];
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
			}
		}
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++)
		s += a[i];
	out[blockIdx.x * 256 + threadIdx.x + ID] = s;
}

typedef void (*kfn)(float *, int, float, float);

template <int UNROLL, int CHAINS>
static void run(const char *label, float *out, hipStream_t *st, int total_insts)
{
	kfn same[4] = { kern<0, UNROLL, CHAINS>, kern<0, UNROLL, CHAINS>, kern<0, UNROLL, CHAINS>, kern<0, UNROLL, CHAINS> };
	kfn diff[4] = { kern<0, UNROLL, CHAINS>, kern<1, UNROLL, CHAINS>, kern<2, UNROLL, CHAINS>, kern<3, UNROLL, CHAINS> };
	const int iters = total_insts / (UNROLL * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto timed = [&](kfn *k, int n) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; rep++) {
			hipDeviceSynchronize();
			hipEventRecord(e0, st[0]);
			for (int i = 1; i < n; i++)
				hipStreamWaitEvent(st[i], e0, 0);
			for (int i = 0; i < n; i++)
				hipLaunchKernelGGL(k[i], dim3(256), dim3(256), 0, st[i], out, iters, 1.0001f, 0.5f);
			hipDeviceSynchronize();
			hipEventRecord(e1, st[0]);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			best = ms < best ? ms : best;
		}
		return best;
	};
	for (int i = 0; i < 4; i++) {  // warm (code load)
		hipLaunchKernelGGL(diff[i], dim3(256), dim3(256), 0, st[0], out, 1, 1.0f, 0.0f);
	}
	const float one = timed(same, 1), s2 = timed(same, 2), d2 = timed(diff, 2), s4 = timed(same, 4), d4 = timed(diff, 4);
	printf("%-10s body %3d KB chains %d: alone %.3f ms | 2 same %.3f  2 different %.3f | 4 same %.3f  4 different %.3f  (x%.2f)\n", label,
	       UNROLL * 8 * 8 / 1024, CHAINS, one, s2, d2, s4, d4, d4 / s4);
}

int main()
{
	float *out;
	hipMalloc(&out, 1 << 22);
	hipStream_t st[4];
	for (int i = 0; i < 4; i++)
		hipStreamCreate(&st[i]);
	const int N = 1 << 21;  // wave instructions per kernel
	run<32, 8>("2KB", out, st, N);
	run<128, 8>("8KB", out, st, N);
	run<256, 8>("16KB", out, st, N);
	run<384, 8>("24KB", out, st, N);
	run<512, 8>("32KB", out, st, N);
	run<768, 8>("48KB", out, st, N);
	run<32, 1>("2KB", out, st, N / 4);
	run<128, 1>("8KB", out, st, N / 4);
	run<256, 1>("16KB", out, st, N / 4);
	run<384, 1>("24KB", out, st, N / 4);
	run<512, 1>("32KB", out, st, N / 4);
	run<768, 1>("48KB", out, st, N / 4);
	return 0;
}
