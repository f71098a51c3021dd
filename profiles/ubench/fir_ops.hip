// Throughput of the two ways to form sum_n floor(x[n]*h[n] / 65536) over 20 taps (front end, stage 2):
//  A: 20 x (v_mul_hi_i32_i24 + v_add)            B: 10 x (v_dot2c_i32_i16 + v_pk_mul_lo_u16 + v_dot2_u32_u16)
// build: hipcc --offload-arch=gfx950 -O3 fir_ops.hip -o fir_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
struct Taps { int t[20]; };
template <int MODE>
__global__ __launch_bounds__(256) void k(const int *in, int *out, Taps taps, int iters)
{
	const int tid = blockIdx.x * 256 + threadIdx.x;
	int y[28];
	for (int i = 0; i < 28; i++) y[i] = in[(tid + i * 7) & 1023];
	int acc = 0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int o = 0; o < 4; o++) {
			if (MODE == 0) {
				int s = 0;
#pragma unroll
				for (int n = 0; n < 20; n++) {
					int r;
					asm("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(r) : "s"(taps.t[n]), "v"(y[2 * o + n]));
					s += r;
				}
				acc += (short)s;
			} else {
				int h = 0; unsigned l = 0;
#pragma unroll
				for (int n = 0; n < 10; n++) {
					const short2v x = __builtin_bit_cast(short2v, y[o + n]), t = __builtin_bit_cast(short2v, taps.t[n]);
					h = __builtin_amdgcn_sdot2(x, t, h, false);
					const ushort2v lo = __builtin_bit_cast(ushort2v, y[o + n]) * __builtin_bit_cast(ushort2v, taps.t[n]);
					l = __builtin_amdgcn_udot2(lo, __builtin_bit_cast(ushort2v, 0x00010001), l, false);
				}
				acc += (short)((h - (int)l) >> 16);
			}
		}
#pragma unroll
		for (int i = 0; i < 28; i++) y[i] += acc + i;
	}
	out[tid] = acc;
}
int main()
{
	int *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096); hipMemset(in, 1, 4096);
	Taps t; for (int i = 0; i < 20; i++) t.t[i] = 1000 + 37 * i;
	for (int mode = 0; mode < 2; mode++) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			if (mode == 0) k<0><<<4096, 256>>>(in, out, t, 2000); else k<1><<<4096, 256>>>(in, out, t, 2000);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("%s: %.3f ms  (%.1f G outputs/s)\n", mode == 0 ? "A mulhi24+add   " : "B dot2c+pkmul+dot2", ms, 4096.0 * 256 * 2000 * 4 / ms / 1e6);
	}
	return 0;
}
