// Micro-benchmark: serial lanes streaming their own far-apart rows (the access pattern of the chain kernels).
// build: hipcc --offload-arch=gfx950 -O3 lane_stream.hip -o lane_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Slot8 { uint4 q0, q1, q2, q3, q4, q5, q6, q7; };
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t *base, size_t row_words, int iters, int lanes, unsigned *out)
{
	__shared__ uint4 lds[8 * 64];
	if ((int)threadIdx.x >= lanes) return;
	const int s = blockIdx.x * lanes + threadIdx.x;
	const uint4 *row = reinterpret_cast<const uint4 *>(base + (size_t)s * row_words);
	uint4 *my = lds + threadIdx.x;
	unsigned acc = 0;
	auto load = [&](int i) { Slot8 r; const uint4 *p = row + 8 * i; r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2]; r.q3 = p[3]; r.q4 = p[4]; r.q5 = p[5]; r.q6 = p[6]; r.q7 = p[7]; return r; };
	Slot8 cur = load(0);
	for (int i = 0; i < iters; i++) {
		Slot8 nxt = load(i + 1 < iters ? i + 1 : i);
		if (MODE == 0) {  // consume registers directly
			acc += cur.q0.x + cur.q1.y + cur.q2.z + cur.q3.w + cur.q4.x + cur.q5.y + cur.q6.z + cur.q7.w;
		} else {          // LDS staging + rolled loop of 8 groups x 4 samples, ~MODE instrs per sample
			my[0] = cur.q0; my[64] = cur.q1; my[128] = cur.q2; my[192] = cur.q3; my[256] = cur.q4; my[320] = cur.q5; my[384] = cur.q6; my[448] = cur.q7;
			uint4 vn = my[0];
#pragma unroll 1
			for (int q = 0; q < 8; q++) {
				uint4 v = vn; vn = my[((q + 1) & 7) * 64];
				unsigned w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int t = 0; t < 4; t++) {
					unsigned x = w[t];
#pragma unroll
					for (int r = 0; r < MODE; r++) x = x * 1664525u + acc;
					acc ^= x;
				}
			}
		}
		cur = nxt;
	}
	out[s] = acc;
}
int main()
{
	const int streams = 1024, iters = 5500; const size_t row_words = 428544;  // 1.71 MB rows like dev32
	uint32_t *d; unsigned *o; hipMalloc(&d, (size_t)streams * row_words * 4); hipMalloc(&o, streams * 4);
	hipMemset(d, 1, (size_t)streams * row_words * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(M, LANES) { k<M><<<(streams + LANES - 1) / LANES, 64>>>(d, row_words, iters, LANES, o); hipDeviceSynchronize(); hipEventRecord(e0); \
	k<M><<<(streams + LANES - 1) / LANES, 64>>>(d, row_words, iters, LANES, o); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
	printf("mode %2d lanes/wave %2d: %.3f ms  (%.1f ns per 32-sample slot per lane, %.0f cycles/sample @2.4GHz)\n", M, LANES, ms, ms * 1e6 / iters, ms * 1e6 / iters / 32 * 2.4); }
	RUN(0, 64) RUN(0, 16) RUN(0, 4) RUN(1, 64) RUN(4, 64) RUN(16, 64) RUN(16, 16) RUN(40, 64)
	return 0;
}
