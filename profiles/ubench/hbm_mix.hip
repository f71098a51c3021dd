// How fast are this project's access patterns, alone and against each other?  (The PMC calibration next door counts
// bytes; this one times.)  Patterns: coalesced 16 B/lane reads and writes (front end, discriminator), lane-per-row 16-byte
// reads / writes 8 KiB apart (the biquad passes' segments).  Then the coalesced read is timed while a lane-per-row kernel
// runs beside it on another stream.
// build: hipcc --offload-arch=gfx950 -O3 hbm_mix.hip -o hbm_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr size_t kBytes = 2ull << 30;
constexpr int kRow = 8192;
__global__ void rd16(const uint4 *p, uint4 *o, size_t n) { uint4 a = {0,0,0,0}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if (a.x == 0x12345678) o[0] = a; }
__global__ void wr16(uint4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void rdrow(const uint4 *p, uint4 *o, size_t rows) { const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (t >= rows) return; uint4 a = {0,0,0,0}; const uint4 *r = p + t * (kRow / 16); for (int i = 0; i < kRow / 16; i++) { uint4 v = r[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if (a.x == 0x12345678) o[0] = a; }
__global__ void wrrow(uint4 *p, size_t rows) { const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (t >= rows) return; uint4 *r = p + t * (kRow / 16); for (int i = 0; i < kRow / 16; i++) r[i] = make_uint4((uint32_t)t, i, 2, 3); }
// lane-per-row write of 64-byte groups: four 16-byte stores back to back (what a full-line store per lane would be)
__global__ void wrrow64(uint4 *p, size_t rows) { const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (t >= rows) return; uint4 *r = p + t * (kRow / 16); for (int i = 0; i < kRow / 16; i += 4) { r[i] = make_uint4((uint32_t)t, i, 2, 3); r[i + 1] = make_uint4((uint32_t)t, i, 2, 4); r[i + 2] = make_uint4((uint32_t)t, i, 2, 5); r[i + 3] = make_uint4((uint32_t)t, i, 2, 6); } }

// the same bytes as wrrow, but each store instruction covers 16 rows x 64 contiguous bytes (PIECES = 4) or 8 rows x 128
// (PIECES = 8): what an LDS transposition of the wave's 64 row pieces gives
template <int PIECES> __global__ void wrrow_t(uint4 *p, size_t rows)
{
	const size_t t0 = blockIdx.x * (size_t)blockDim.x;  // the wave's first row
	if (t0 >= rows) return;
	const int ln = threadIdx.x, piece = ln % PIECES, rsub = ln / PIECES;
	constexpr int RPI = 64 / PIECES;
	for (int i = 0; i < kRow / 16; i += PIECES)      // column group of PIECES x 16 bytes
		for (int rg = 0; rg < 64; rg += RPI) {      // row group
			uint4 *r = p + (t0 + rg + rsub) * (kRow / 16);
			r[i + piece] = make_uint4((uint32_t)t0, i, 2, 3);
		}
}
template <class F> static float timed(hipStream_t st, F f)
{
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	f(); hipStreamSynchronize(st);
	hipEventRecord(a, st); f(); hipEventRecord(b, st); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
	void *b1, *b2, *o;
	if (hipMalloc(&b1, kBytes) != hipSuccess || hipMalloc(&b2, kBytes) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) return 1;
	hipMemset(b1, 1, kBytes); hipMemset(b2, 1, kBytes);
	hipStream_t s1, s2;
	hipStreamCreate(&s1); hipStreamCreate(&s2);
	const int g = 256 * 16;
	const unsigned gr = (unsigned)(kBytes / kRow / 64);
	auto k_rd16 = [&] { rd16<<<g, 256, 0, s1>>>((const uint4 *)b1, (uint4 *)o, kBytes / 16); };
	auto k_wr16 = [&] { wr16<<<g, 256, 0, s1>>>((uint4 *)b1, kBytes / 16); };
	auto k_rdrow = [&] { rdrow<<<gr, 64, 0, s1>>>((const uint4 *)b1, (uint4 *)o, kBytes / kRow); };
	auto k_wrrow = [&] { wrrow<<<gr, 64, 0, s1>>>((uint4 *)b1, kBytes / kRow); };
	auto k_wrrow64 = [&] { wrrow64<<<gr, 64, 0, s1>>>((uint4 *)b1, kBytes / kRow); };
	auto k_wrt4 = [&] { wrrow_t<4><<<gr, 64, 0, s1>>>((uint4 *)b1, kBytes / kRow); };
	auto k_wrt8 = [&] { wrrow_t<8><<<gr, 64, 0, s1>>>((uint4 *)b1, kBytes / kRow); };
	const double gb = kBytes / 1e9;
	printf("alone: rd16 %.0f GB/s, wr16 %.0f, rdrow %.0f, wrrow %.0f, wrrow64 %.0f\n", gb / timed(s1, k_rd16) * 1e3, gb / timed(s1, k_wr16) * 1e3,
	       gb / timed(s1, k_rdrow) * 1e3, gb / timed(s1, k_wrrow) * 1e3, gb / timed(s1, k_wrrow64) * 1e3);
	printf("alone: wrrow_t<4> (64-byte pieces) %.0f GB/s, wrrow_t<8> (128-byte pieces) %.0f\n", gb / timed(s1, k_wrt4) * 1e3, gb / timed(s1, k_wrt8) * 1e3);
	// the coalesced read while a lane-per-row kernel runs beside it (launched 6x so it outlasts the read)
	for (int which = 0; which < 5; which++) {
		for (int r = 0; r < 6; r++) {
			if (which == 0) rdrow<<<gr, 64, 0, s2>>>((const uint4 *)b2, (uint4 *)o, kBytes / kRow);
			else if (which == 1) wrrow<<<gr, 64, 0, s2>>>((uint4 *)b2, kBytes / kRow);
			else if (which == 2) wr16<<<g, 256, 0, s2>>>((uint4 *)b2, kBytes / 16);
			else if (which == 3) wrrow_t<4><<<gr, 64, 0, s2>>>((uint4 *)b2, kBytes / kRow);
			else wrrow_t<8><<<gr, 64, 0, s2>>>((uint4 *)b2, kBytes / kRow);
		}
		hipEvent_t a, b;
		hipEventCreate(&a); hipEventCreate(&b);
		hipEventRecord(a, s1); k_rd16(); hipEventRecord(b, s1); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		hipEvent_t c, d; hipEventCreate(&c); hipEventCreate(&d);
		hipDeviceSynchronize();
		printf("rd16 beside %s: %.0f GB/s\n", which == 0 ? "rdrow" : which == 1 ? "wrrow" : which == 2 ? "wr16" : which == 3 ? "wrrow_t<4>" : "wrrow_t<8>", gb / ms * 1e3);
	}
	return 0;
}
