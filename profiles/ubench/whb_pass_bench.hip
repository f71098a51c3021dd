// Micro-benchmark: the WHB stage-2 "whole slot, unsynced" pass (decision-level biquad + candidate mask) on
// LDS-resident data, exactly as in chains2.hip, timed with clock64 for 1 wave per block.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../tfrec_amd/csrc whb_pass_bench.hip -o whb_pass_bench
#include <stdio.h>
#include "dsp_dev.h"
using namespace tfrec;
struct WhbFast { Biquad iir_avg; int avg_of, last_dev; };
template <int MODE>
__global__ __launch_bounds__(64) void k(long long *cyc, unsigned *out, BiquadCoef c, int iters)
{
	__shared__ uint4 lds[8 * 64];
	uint4 *my = lds + threadIdx.x;
	for (int q = 0; q < 8; q++) my[q * 64] = make_uint4(1000 * q + threadIdx.x, -300 * q, 77 * q, 5000 - q);
	WhbFast w{ { 0, 0, 0, 0 }, 0, 0 };
	unsigned acc = 0;
	long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
		uint32_t mask = 0;
		if (MODE == 0) {  // unsynced: biquad + candidates
			Biquad f = w.iir_avg; int avg = w.avg_of, last = w.last_dev;
			uint4 vn = my[0];
#pragma unroll 1
			for (int q = 0; q < 8; q++) {
				const uint4 v = vn; vn = my[((q + 1) & 7) * 64];
				const uint32_t dv[4] = { v.x, v.y, v.z, v.w }; uint32_t m4 = 0;
#pragma unroll
				for (int t = 0; t < 4; t++) { const int dev = (int)dv[t]; avg = (int)iir_step(f, c, 0.5 * (double)dev); m4 |= (uint32_t)(dev < avg && dev > last) << t; last = dev; }
				mask |= m4 << (4 * q);
			}
			w.iir_avg = f; w.avg_of = avg; w.last_dev = last;
		} else {  // synced: candidates only
			const int avg = w.avg_of + it; int last = w.last_dev;
#pragma unroll
			for (int q = 0; q < 8; q++) { const uint4 v = my[q * 64]; const uint32_t dv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int t = 0; t < 4; t++) { const int dev = (int)dv[t]; mask |= (uint32_t)(dev < avg && dev > last) << (4 * q + t); last = dev; } }
			w.last_dev = last;
		}
		acc ^= mask;
		my[(it & 7) * 64].x += acc & 3;  // keep the data changing
	}
	long long t1 = clock64();
	out[blockIdx.x * 64 + threadIdx.x] = acc + (unsigned)w.avg_of;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
	long long *c; unsigned *o; hipMalloc(&c, 8 * 64); hipMalloc(&o, 4 * 64 * 64);
	BiquadCoef cf = { 0x1.02ae4cfc8910ap-26, 0x1.02ae4cfc8910ap-25, 0x1.02ae4cfc8910ap-26, 0x1.ffe9409fe171bp+0, -0x1.ffd283451f7d3p-1 };
	const int iters = 2000; long long h;
	k<0><<<1, 64>>>(c, o, cf, iters); hipDeviceSynchronize(); k<0><<<1, 64>>>(c, o, cf, iters); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
	printf("unsynced pass: %.1f cycles per sample (%.0f per 32-sample slot)\n", (double)h / iters / 32, (double)h / iters);
	k<1><<<1, 64>>>(c, o, cf, iters); hipDeviceSynchronize(); k<1><<<1, 64>>>(c, o, cf, iters); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
	printf("synced pass:   %.1f cycles per sample (%.0f per 32-sample slot)\n", (double)h / iters / 32, (double)h / iters);
	return 0;
}
