// Known-byte kernels in the access patterns this project uses, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: only the 16 B/lane coalesced read is calibrated there: x2).  Every kernel moves
// exactly N bytes of a 2 GiB buffer (past the 256 MiB Infinity Cache).  profiles/run_calib.sh runs it under two PMC passes
// and profiles/make_calibration.py writes profiles/r02_hbm_calibration.json.
// build: hipcc --offload-arch=gfx950 -O3 hbm_calib.hip -o hbm_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr size_t kBytes = 2ull << 30;
constexpr int kRow = 8192;  // bytes between the rows of the lane-per-row patterns (a biquad segment: 4096 int16)
__global__ void rd16(const uint4 *p, uint4 *o, size_t n) { uint4 a = {0,0,0,0}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if (a.x == 0x12345678) o[0] = a; }
__global__ void rd4(const uint32_t *p, uint32_t *o, size_t n) { uint32_t a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i]; if (a == 0x12345678) o[0] = a; }
__global__ void rd2(const uint16_t *p, uint16_t *o, size_t n) { uint32_t a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i]; if (a == 0x1234) o[0] = (uint16_t)a; }
// lane per row: thread t streams row t (kRow bytes) with 16-byte loads, like spec_biquad_kernel's k3_load
__global__ void rdrow(const uint4 *p, uint4 *o, size_t rows) { const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (t >= rows) return; uint4 a = {0,0,0,0}; const uint4 *r = p + t * (kRow / 16); for (int i = 0; i < kRow / 16; i++) { uint4 v = r[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if (a.x == 0x12345678) o[0] = a; }
__global__ void wr16(uint4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void wr4(uint32_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
__global__ void wr2(uint16_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint16_t)i; }
__global__ void wrrow(uint4 *p, size_t rows) { const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (t >= rows) return; uint4 *r = p + t * (kRow / 16); for (int i = 0; i < kRow / 16; i++) r[i] = make_uint4((uint32_t)t, i, 2, 3); }
int main()
{
	void *b, *o;
	if (hipMalloc(&b, kBytes) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) return 1;
	hipMemset(b, 1, kBytes);
	const int g = 256 * 16;
	for (int rep = 0; rep < 2; rep++) {
		rd16<<<g, 256>>>((const uint4 *)b, (uint4 *)o, kBytes / 16);
		rd4<<<g, 256>>>((const uint32_t *)b, (uint32_t *)o, kBytes / 4);
		rd2<<<g, 256>>>((const uint16_t *)b, (uint16_t *)o, kBytes / 2);
		rdrow<<<(unsigned)(kBytes / kRow / 64), 64>>>((const uint4 *)b, (uint4 *)o, kBytes / kRow);
		wr16<<<g, 256>>>((uint4 *)b, kBytes / 16);
		wr4<<<g, 256>>>((uint32_t *)b, kBytes / 4);
		wr2<<<g, 256>>>((uint16_t *)b, kBytes / 2);
		wrrow<<<(unsigned)(kBytes / kRow / 64), 64>>>((uint4 *)b, kBytes / kRow);
	}
	hipDeviceSynchronize();
	printf("each kernel moves %zu bytes\n", kBytes);
	return 0;
}
