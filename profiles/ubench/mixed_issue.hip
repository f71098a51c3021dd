// profiles/ubench/mixed_issue.hip -- can a SIMD issue a vector and a scalar instruction of two DIFFERENT waves in the same
// quad-cycle?  (the batch period equals the sum of SQ_ACTIVE_INST_ANY over the SIMDs -- is that a hardware
// limit of one instruction per SIMD and quad-cycle, or a consequence of too few ready waves?)
// W waves per SIMD.  Mixed: workgroups of 512 threads, waves 0-3 run a stream of independent v_fma_f64, waves 4-7 (the
// same four SIMDs) a stream of s_add_u32; "valu" / "salu": all workgroups the same class, W/2 per CU --
// the same work per class as in the mixed run.  If the classes overlap, t(mixed) ~ max(t(valu), t(salu)); if a SIMD issues
// one instruction of any kind per quad-cycle, t(mixed) ~ t(valu) + t(salu).
// Also "one wave, alternating": v_fma_f64 and s_add_u32 alternate in ONE wave's instruction stream.
// build: hipcc --offload-arch=gfx950 -O2 -o mixed_issue mixed_issue.hip ; run: ./mixed_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// mode 0: every block VALU; 1: every block SALU; 2: even blocks VALU, odd SALU; 3: alternating in one wave;
// 4: even blocks v_fma_f64, odd blocks v_pk_fma_f32 (two vector classes: must add up)
__global__ __launch_bounds__(512) void k(int mode, int reps, float *sink)
{
	double d[8];
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p[8];
	for (int i = 0; i < 8; i++) {
		d[i] = threadIdx.x * 1e-3 + i;
		p[i] = f2{ (float)d[i], (float)d[i] + 1 };
	}
	const double cd = 1.0001;
	const f2 cp = { 1.0001f, 0.9999f };
	int s0 = reps, s1 = 3;
	// (512-thread workgroups in the mixed modes: waves w and w + 4 of a workgroup share a SIMD of its CU)
	const bool odd = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) & 1;
	const int cls = mode == 0 ? 0 : (mode == 1 ? 1 : (mode == 2 ? (odd ? 1 : 0) : (mode == 3 ? 2 : (odd ? 3 : 0))));
	for (int r = 0; r < reps; r++) {
		if (cls == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
			REP64(X)
#undef X
		} else if (cls == 1) {
#define X(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
			REP64(X)
#undef X
		} else if (cls == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %2\n s_add_u32 %1, %1, %3" : "+v"(d[i]), "+s"(s0) : "v"(cd), "s"(s1) : "scc");
			REP64(X)
#undef X
		} else {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
			REP64(X)
#undef X
		}
	}
	float acc = 0;
	for (int i = 0; i < 8; i++)
		acc += (float)d[i] + p[i].x + p[i].y;
	acc += s0;
	if (acc == 12345.678f)
		*sink = acc;
}

static float run(int mode, int blocks, int reps, float *sink, int threads = 256)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, 10, sink);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, reps, sink);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main()
{
	hipDeviceProp_t pr;
	hipGetDeviceProperties(&pr, 0);
	const int n_cu = pr.multiProcessorCount;
	float *sink;
	hipMalloc(&sink, 4);
	const int reps = 2000;
	for (int W : { 2, 4, 8 }) {
		const float tv = run(0, n_cu * W / 2, reps, sink), ts = run(1, n_cu * W / 2, reps, sink);
		const float tm = run(2, n_cu * W / 2, reps, sink, 512), tvv = run(4, n_cu * W / 2, reps, sink, 512);
		const float tpk = run(0, n_cu * W, reps, sink);
		printf("{\"waves_per_simd\": %d, \"valu_only_ms\": %.3f, \"salu_only_ms\": %.3f, \"mixed_valu_salu_ms\": %.3f, "
		       "\"sum_ms\": %.3f, \"max_ms\": %.3f, \"two_vector_classes_ms\": %.3f, \"valu_all_blocks_ms\": %.3f}\n",
		       W, tv, ts, tm, tv + ts, tv > ts ? tv : ts, tvv, tpk);
	}
	const float ta = run(3, n_cu, reps, sink), t1v = run(0, n_cu, reps, sink), t1s = run(1, n_cu, reps, sink);
	printf("{\"one_wave_per_simd\": 1, \"alternating_ms\": %.3f, \"valu_only_ms\": %.3f, \"salu_only_ms\": %.3f}\n", ta, t1v, t1s);
	return 0;
}
