// tfrec_amd/csrc/frontend.hip -- front-end kernel: raw u8 IQ -> decimated int16 IQ + trigger mask.
//
// Replaces, for a whole batch of streams, the per-block host work of the reference:
//   engine.cpp:77-78      s16 = (u8 - 128) << 6
//   dsp_stuff.cpp:204-230 decimate::process2x1  (8 taps, 2:1, per-tap arithmetic >>16, int16 store)
//   dsp_stuff.cpp:172-202 decimate::process2x   (20 taps, 2:1, same arithmetic; narrow or -W wide taps)
//   dsp_stuff.cpp:243-264 downconvert::process_iq (I and Q rails, passes = 2)
//   fm_demod.cpp:45       pwr = abs(I) + abs(Q), compared with the trigger threshold (tfa1.cpp:147 etc.)
//
// Exact integer semantics: with x = (u8-128)<<6,
//   y1[k] = int16( sum_{n<8}  (x[2k-6+n]  * h1[n]) >> 16 )
//   y2[m] = int16( sum_{n<20} (y1[2m-18+n] * h2[n]) >> 16 )
// The per-tap floor shift forbids folding the symmetric taps or summing before shifting.  Both stages evaluate it as
// ONE fp32 fused multiply-add per tap with the wave's fp32 rounding mode set to "toward -inf":
//     acc = fma(x, h / 2^16, acc),   acc an integer in [2^23, 2^24) (one ulp = 1),  x and h / 2^16 exact in fp32
// The FMA forms x*h/2^16 + acc exactly and rounds ONCE, downwards, to a multiple of acc's ulp: acc + floor(x*h / 2^16),
// i.e. acc + ((x*h) >> 16) -- whatever the width of x*h.  The reference's int16 store of the sum is the low half of
// acc's mantissa field (the starting value 2^23 + 2^22 has no bits there).  v_pk_fma_f32 does the I and the Q rail at
// once: one VALU instruction per tap and complex sample, where the integer form (v_mul_hi_i32_i24 + add) took 3.5.
// Stage 1 with u8 input uses ((u8-128)*h) >> 10 (identical value, the <<6 cancels).
//
// MI355X mapping: one 256-thread workgroup per tile of 2048 decimated outputs of one stream (kFrontOut = 8 per thread).
// A lane reads the 76 raw bytes of a stage-1 group (16 outputs of both rails) straight from global memory with five
// unaligned vector loads -- neighbouring lanes overlap by 12 bytes (64-byte stride), so the tile streams its 16 KiB +
// 112 B halo once (the halo of a submit's first tile comes from the previous submit's tail); stage-1 outputs live only
// in LDS ((I, Q) float pairs, 37 KB per workgroup, a pad of 16 bytes behind every lane's 128), read back with
// ds_read_b128; stage-2 results leave as two 16-byte stores per thread, and the trigger bits of 64 consecutive samples
// are packed into one 64-bit word per 8 lanes with DPP.  No MFMA: the path is a streaming stencil with a per-tap
// rounding, bound by HBM bytes and VALU issue.  (Round 2-4 made 4 outputs per thread: the same FMAs, but 16 % more byte
// conversions -- the overlap is per group --, twice the scalar and address work per output: profiles/NOTES.md round 5.)
#include <stdlib.h>

#include "dsp_dev.h"
#include "knobs.h"
#include <algorithm>

namespace tfrec {

__device__ __constant__ double kAtanPolyFront[16] = TFREC_ATAN_POLY;  // see dsp_dev.h

// first-stage taps (dsp_stuff.cpp:119-130)
__device__ __constant__ const int kS1[8] = { 2443, 6339, 11036, 14254, 14254, 11036, 6339, 2443 };

typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load, dword aligned

// IN16 = false: raw input is u8 IQ, x = (u8 - 128) << 6 (engine.cpp:77-78).  IN16 = true: the input already is
// int16 (I,Q) pairs at 1.536 MS/s (what decim10_kernel produces for BASELINE config 5): 4 bytes per complex
// sample instead of 2, per-tap (x*h)>>16 without the <<6 shortcut.
// The front end runs as kFrontPersist PERSISTENT workgroups that take the tiles in turn (round 6).  As a workgroup per tile its
// grid is 196 k workgroups on the context's highest-priority stream: they hold the workgroup dispatcher until the last one is
// placed, and for those ~2 ms of every period no kernel of another stream STARTS (profiles/r06_final_steps.txt: everything
// begins in the moment the front end ends).  At lower priority the front end starves instead (rounds 4 and 6).  With a few
// workgroups per CU placed at once its queue is empty, the others start beside it: -3 % (profiles/r06_ab_front_end_dispatch.txt).
constexpr int kFrontPersist = 2048;
template <bool IN16>
__global__ __launch_bounds__(kFrontThreads) void frontend_kernel(
	const uint8_t *__restrict__ iq, size_t stride, int m_total, const uint8_t *__restrict__ tail_in,
	uint8_t *__restrict__ tail_out, uint32_t *__restrict__ dec, size_t dec_stride,
	unsigned long long *__restrict__ mask, size_t mask_stride, uint32_t *__restrict__ prevdec, int thresh, FrontTaps taps,
	int n_streams, int persist)
{
	constexpr int kB = IN16 ? 2 : 1;             // bytes per rail sample
	constexpr int kTail = kTailBytes * kB;      // history bytes (56 complex samples)
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	typedef float f32x4 __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) f32x2 y1[kY1Count];  // stage-1 outputs as (I, Q) pairs of floats
	const float kMagic = 12582912.0f;  // 2^23 + 2^22, see stage 1
	const long nbytes = 8L * kB * m_total;
	// MODE.FP_ROUND (fp32) = 2, round toward -inf: see stage 1.  Every fp32 operation of this kernel is either one of
	// those FMAs or exact.
	__builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 2);
#ifdef TFREC_AMD_FE_PRIO
	__builtin_amdgcn_s_setprio(TFREC_AMD_FE_PRIO);
#endif
	// A workgroup per tile (persist = 0: grid = tiles x streams), or -- TFREC_AMD_FE_PERSIST=n -- n workgroups that take the
	// tiles in turn: the stream's queue is then empty as soon as they are placed, and other streams' kernels can START beside a
	// running front end (its 196 k workgroups at high priority hold the dispatcher for their whole duration otherwise)
	const int ntiles = m_total / kTileDec;
	const int n_work = persist ? ntiles * n_streams : 1;
	for (int w = persist ? (int)blockIdx.x : 0; w < n_work; w += persist ? (int)gridDim.x : 1) {
	const int s = persist ? w / ntiles : (int)blockIdx.y;
	const int tile = persist ? w - s * ntiles : (int)blockIdx.x;
	// (the lane index laundered per tile: what the tile derives from it is computed again instead of being kept alive across the
	// whole loop -- with everything hoisted the kernel needed 153 registers instead of 97, three waves per SIMD instead of five)
	int tid = threadIdx.x;
	asm volatile("" : "+v"(tid));
	const int m0 = tile * kTileDec;
	const uint8_t *src = iq + (size_t)s * stride;
	// The tile reads raw bytes [8*m0 - 112, 8*m0 + 8*T + 16) (x kB) straight from global memory: a lane's 76 bytes per
	// stage-1 group overlap its neighbours' (64-byte stride), so the last load of a group hits what the first one of
	// the next lane brought in; the 112 bytes before the submit come from the previous one's tail, what lies behind its
	// end is silence (only the last tile's last groups look there, and their outputs are never used).
	const long base = 8L * kB * m0 - kTail;
	const bool interior = tile > 0 && tile + 1 < ntiles;
	const uint32_t silence = IN16 ? 0u : 0x80808080u;
	auto raw_dword = [&](long bo) -> uint32_t {  // bo: byte offset into the stream, 4-byte aligned; edges only
		if (bo >= 0 && bo + 4 <= nbytes)
			return *reinterpret_cast<const uint32_t *>(src + bo);
		if (bo < 0)
			return *reinterpret_cast<const uint32_t *>(tail_in + (size_t)s * kTail + (kTail + bo));
		return silence;
	};
	// history for the next submit: the last 56 raw complex samples of this one
	if (tile == ntiles - 1 && tid < kTail / 16)
		*reinterpret_cast<uint4 *>(tail_out + (size_t)s * kTail + 16 * tid) =
			*reinterpret_cast<const uint4 *>(src + nbytes - kTail + 16 * tid);

	// ---- stage 1: LDS slot i <-> y1[2*m0 - 22 + i] (slot i at y1[y1_phys(i)]); a lane makes kG1 = 2 * kFrontOut consecutive
	// outputs of both rails (one pass of the 256 lanes makes the tile's 2*T).  Outputs k..k+kG1-1 need x[2k-6 .. 2k+2*kG1-1]:
	// 4*kG1 + 12 raw bytes at offset base + 12 + 4*kG1*grp (k = 2*m0 - 22 + kG1*grp).  The tile needs 2*T + 24: the last 24
	// one per lane (below).
	constexpr int kG1 = kY1Group;
	constexpr int kDw1 = (2 * kG1 + 6) * kB / 2;  // raw dwords per group: 19 (u8) / 38 (int16) for 16 outputs, 11 / 22 for 8
	static_assert((2 * kTileDec) / kG1 == kFrontThreads, "one full pass");
	typedef uint32_t u32x3_u __attribute__((ext_vector_type(3), aligned(4)));
	typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
	{
		const int grp = tid;
		uint32_t rp[kDw1];
		{
			const int off = kB * (12 + 4 * kG1 * grp);  // from `base`
			// every tile but the submit's first and last reads inside the stream: no bounds to check (a wave-uniform branch;
			// the 64-bit compares of the checked path were a tenth of the kernel's vector instructions)
			if (interior || (base + off >= 0 && base + off + 4 * kDw1 <= nbytes)) {
				const uint8_t *gp = src + base + off;
				const u32x4_u *g = reinterpret_cast<const u32x4_u *>(gp);
#pragma unroll
				for (int q = 0; q < kDw1 / 4; q++) {
					const u32x4_u v = g[q];
					rp[4 * q] = v.x; rp[4 * q + 1] = v.y; rp[4 * q + 2] = v.z; rp[4 * q + 3] = v.w;
				}
				constexpr int kRest = kDw1 % 4, kDone = kDw1 - kRest;
				if constexpr (kRest == 3) {
					const u32x3_u v = *reinterpret_cast<const u32x3_u *>(gp + 4 * kDone);
					rp[kDone] = v.x; rp[kDone + 1] = v.y; rp[kDone + 2] = v.z;
				} else if constexpr (kRest == 2) {
					const u32x2_u v = *reinterpret_cast<const u32x2_u *>(gp + 4 * kDone);
					rp[kDone] = v.x; rp[kDone + 1] = v.y;
				}
				static_assert(kRest == 3 || kRest == 2, "19 / 38 or 11 / 22 dwords");
			} else {
#pragma unroll
				for (int q = 0; q < kDw1; q++)
					rp[q] = raw_dword(base + off + 4 * q);
			}
		}
		f32x2 oy[kG1];
		if (IN16) {
			// int16 input: the same FMA form (below) with x * (h / 65536); x * h has up to 30 bits, which the FMA does not
			// care about (it rounds once, after the exact product), the 8 terms sum to less than 2^16 in magnitude, and
			// the int16 store of the reference (dsp_stuff.cpp:222) -- it can wrap here -- is the low half of the mantissa
			f32x2 x[2 * kG1 + 6];
#pragma unroll
			for (int i = 0; i < 2 * kG1 + 6; i++)
				x[i] = f32x2{ (float)(int)(int16_t)(rp[i] & 0xffff), (float)((int)rp[i] >> 16) };
			f32x2 acc[kG1];
#pragma unroll
			for (int n = 0; n < 8; n++) {
				const float hs = (float)kS1[n] * (1.0f / 65536.0f);
#pragma unroll
				for (int o = 0; o < kG1; o++)
					acc[o] = __builtin_elementwise_fma(x[2 * o + n], f32x2{ hs, hs }, n == 0 ? f32x2{ kMagic, kMagic } : acc[o]);
				__builtin_amdgcn_sched_barrier(0);  // (a tap's FMAs on different accumulators back to back: no wait states)
			}
#pragma unroll
			for (int o = 0; o < kG1; o++)
				oy[o] = f32x2{ (float)(int)(int16_t)(__float_as_uint(acc[o].x) & 0xffffu),
					       (float)(int)(int16_t)(__float_as_uint(acc[o].y) & 0xffffu) };
		} else {
			// u8 input: d = u8 - 128 has 8 bits and h 14, so d * (h / 1024) is exact in fp32, and with the wave's fp32
			// rounding mode set to "toward -inf" (top of the kernel)
			//     acc = fma(d, h / 1024, acc),  acc an integer in [2^23, 2^24)  (one ulp = 1)
			// adds exactly floor(d * h / 1024) = (d * h) >> 10 to acc: the per-tap arithmetic shift of the reference in
			// ONE instruction per tap -- and v_pk_fma_f32 does the I and the Q rail at once.  (The integer form costs a
			// multiply and an add per tap and rail.)  The 8 taps sum to at most 8 * 1782, so acc stays in range and the
			// int16 store of the reference (dsp_stuff.cpp:222) changes nothing.
			f32x2 d[2 * kG1 + 6];
#pragma unroll
			for (int i = 0; i < kDw1; i++) {
				const uint32_t w = rp[i] ^ 0x80808080u;  // bytes become two's complement: one signed byte conversion each
				d[2 * i] = f32x2{ (float)(signed char)(w), (float)(signed char)(w >> 8) };
				d[2 * i + 1] = f32x2{ (float)(signed char)(w >> 16), (float)((int)w >> 24) };
			}
			f32x2 acc[kG1];
#pragma unroll
			for (int n = 0; n < 8; n++) {
				const float hs = (float)kS1[n] * (1.0f / 1024.0f);
#pragma unroll
				for (int o = 0; o < kG1; o++)  // (2^23 + 2^22: room for negative sums)
					acc[o] = __builtin_elementwise_fma(d[2 * o + n], f32x2{ hs, hs }, n == 0 ? f32x2{ kMagic, kMagic } : acc[o]);
				__builtin_amdgcn_sched_barrier(0);  // (a tap's FMAs on different accumulators back to back: no wait states)
			}
#pragma unroll
			for (int o = 0; o < kG1; o++)
				oy[o] = acc[o] - f32x2{ kMagic, kMagic };  // exact
		}
#pragma unroll
		for (int q = 0; q < kG1 / 2; q++)
			*reinterpret_cast<f32x4 *>(&y1[kY1Stride * grp + 2 * q]) = f32x4{ oy[2 * q].x, oy[2 * q].y, oy[2 * q + 1].x, oy[2 * q + 1].y };
	}
	// ---- ... and the 24 stage-1 outputs behind them (slots 2*T .. 2*T + 23: the far end of the tile's last stage-2
	// windows), ONE per lane of the first 24: 8 samples = 16 raw bytes (x kB) at offset 4*slot + 12.  As a third pass of the
	// group loop they cost wave 0 a whole group iteration for 6 busy lanes.
	if (tid < 24) {
		const int slot = 2 * kTileDec + tid;
		const int off = kB * (4 * slot + 12);
		uint32_t rp[4 * kB];
		if (interior || (base + off >= 0 && base + off + 16 * kB <= nbytes)) {
			const u32x4_u *g = reinterpret_cast<const u32x4_u *>(src + base + off);
#pragma unroll
			for (int q = 0; q < kB; q++) {
				const u32x4_u v = g[q];
				rp[4 * q] = v.x; rp[4 * q + 1] = v.y; rp[4 * q + 2] = v.z; rp[4 * q + 3] = v.w;
			}
		} else {
#pragma unroll
			for (int q = 0; q < 4 * kB; q++)
				rp[q] = raw_dword(base + off + 4 * q);
		}
		f32x2 acc = { kMagic, kMagic };
		if (IN16) {
#pragma unroll
			for (int n = 0; n < 8; n++) {
				const float hs = (float)kS1[n] * (1.0f / 65536.0f);
				acc = __builtin_elementwise_fma(f32x2{ (float)(int)(int16_t)(rp[n] & 0xffff), (float)((int)rp[n] >> 16) }, f32x2{ hs, hs }, acc);
			}
			y1[y1_phys(slot)] = f32x2{ (float)(int)(int16_t)(__float_as_uint(acc.x) & 0xffffu), (float)(int)(int16_t)(__float_as_uint(acc.y) & 0xffffu) };
		} else {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t w = rp[i] ^ 0x80808080u;
				const float h0 = (float)kS1[2 * i] * (1.0f / 1024.0f), h1 = (float)kS1[2 * i + 1] * (1.0f / 1024.0f);
				acc = __builtin_elementwise_fma(f32x2{ (float)(signed char)(w), (float)(signed char)(w >> 8) }, f32x2{ h0, h0 }, acc);
				acc = __builtin_elementwise_fma(f32x2{ (float)(signed char)(w >> 16), (float)((int)w >> 24) }, f32x2{ h1, h1 }, acc);
			}
			y1[y1_phys(slot)] = acc - f32x2{ kMagic, kMagic };
		}
	}
	__syncthreads();

	// ---- stage 2: lane makes outputs m0 + R*tid + {0..R-1} (R = kFrontOut); output o needs LDS slots [2*R*tid + 4 + 2*o, +20).
	// Same form as stage 1: acc = fma(y, h / 65536, acc) adds exactly floor(y * h / 65536) = (y * h) >> 16 -- the FMA
	// rounds once, after the exact product, so it does not matter that y * h (up to 31 bits) is not an fp32 number; the
	// 20 per-tap terms sum to less than 2^16, so acc stays in [2^23, 2^24).  One v_pk_fma_f32 per tap for both rails
	// (the integer form: v_mul_hi_i32_i24 + an add per tap and rail).  The int16 store of the reference
	// (dsp_stuff.cpp:196) is the low half of acc's mantissa: the magic constant has no bits there.
	constexpr int R = kFrontOut;
	f32x2 y[2 * R + 18];
#pragma unroll
	for (int i = 0; i < R + 9; i++) {
		const f32x4 a = *reinterpret_cast<const f32x4 *>(&y1[kY1Stride * tid + y1_phys(4 + 2 * i)]);
		y[2 * i] = f32x2{ a.x, a.y };
		y[2 * i + 1] = f32x2{ a.z, a.w };
	}
	uint32_t outw[R];
	uint32_t nib = 0;  // trigger bits of the lane's R samples
	// u8 input: the sums stay below 2^14 in magnitude (|y1| <= 8526, |y2| <= 12153 with the wide taps), so the int16 store
	// changes nothing and acc - 2^23 - 2^22 IS the sample: |I| + |Q| > thresh (fm_demod.cpp:45, tfa1.cpp:147) is evaluated on
	// the floats -- exact integers below 2^17 --, as the sign of (thresh + 0.5) - (|I| + |Q|) (never zero: x - x would be -0
	// in this kernel's rounding mode).  5 instructions per sample instead of 9 and no compare / select pairs with their
	// wait states; the 16-bit halves are packed by one v_perm_b32.
	const float thresh_h = (float)thresh + 0.5f;
	f32x2 accs[R];
#pragma unroll
	for (int o = 0; o < R; o++)
		accs[o] = f32x2{ kMagic, kMagic };
#pragma unroll
	for (int n = 0; n < 20; n++) {
#pragma unroll
		for (int o = 0; o < R; o++)
			accs[o] = __builtin_elementwise_fma(y[2 * o + n], f32x2{ taps.f2[n][0], taps.f2[n][1] }, accs[o]);
		// a tap's FMAs go to R accumulators: none waits for the one before it (chained per accumulator, a packed FMA
		// needs a wait state before its successor)
		__builtin_amdgcn_sched_barrier(0);
	}
#pragma unroll
	for (int o = 0; o < R; o++) {
		const f32x2 acc = accs[o];
		if (IN16) {  // int16 input: the reference's int16 store can wrap (dsp_stuff.cpp:196): integer path
			const int oI = (int)(int16_t)(__float_as_uint(acc.x) & 0xffffu), oQ = (int)(int16_t)(__float_as_uint(acc.y) & 0xffffu);
			outw[o] = ((uint32_t)oI & 0xffffu) | ((uint32_t)oQ << 16);
			nib |= (uint32_t)((abs(oI) + abs(oQ)) > thresh) << o;
		} else {
			const f32x2 d = acc - f32x2{ kMagic, kMagic };
			const float r = thresh_h - (__builtin_fabsf(d.x) + __builtin_fabsf(d.y));
			nib |= (__float_as_uint(r) >> 31) << o;
			outw[o] = __builtin_amdgcn_perm(__float_as_uint(acc.y), __float_as_uint(acc.x), 0x05040100u);
		}
	}
#pragma unroll
	for (int q = 0; q < R / 4; q++)
		*reinterpret_cast<uint4 *>(dec + (size_t)s * dec_stride + m0 + R * tid + 4 * q) =
			make_uint4(outw[4 * q], outw[4 * q + 1], outw[4 * q + 2], outw[4 * q + 3]);

	// ---- trigger mask: bit b of word w <-> decimated sample 64*w + b.  Lane l holds samples R*l .. R*l+R-1 of its wave's
	// 64*R.  R = 4: a nibble at bit 4*(l & 15) of word l >> 4; lanes 0-7 of a row of 16 fill the word's low dword, lanes
	// 8-15 its high dword: shift the nibble within a dword, OR over the 8 lanes with three DPP steps, fetch the other half.
	// R = 8: a byte at bit 8*(l & 7) of word l >> 3; a quad fills a dword (two DPP steps), the next quad is the high half.
	// (Built from four ballots with bit-spreading arithmetic this was a quarter of the kernel's vector instructions.)
	{
		const int lane = tid & 63, wave = tid >> 6;
		if (R == 4) {
			uint32_t v = nib << (4 * (lane & 7));
			v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
			v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
			v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror: the other quad of the 8
			const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8: lane + 8's
			if ((lane & 15) == 0)
				mask[(size_t)s * mask_stride + (m0 >> 6) + 4 * wave + (lane >> 4)] = ((unsigned long long)hi << 32) | v;
		} else {
			uint32_t v = nib << (8 * (lane & 3));
			v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
			v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
			const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);  // row_shl:4: lane + 4's
			if ((lane & 7) == 0)
				mask[(size_t)s * mask_stride + (m0 >> 6) + 8 * wave + (lane >> 3)] = ((unsigned long long)hi << 32) | v;
		}
	}

	// ---- the decimated sample BEFORE this submit's first one (it comes out of the carried raw history; zero history
	// at stream start): the FM discriminator pass needs it for sample 0
	if (tile == 0 && tid == 0) {
		f32x2 acc = { kMagic, kMagic };
#pragma unroll
		for (int n = 0; n < 20; n++)
			acc = __builtin_elementwise_fma(y1[y1_phys(2 + n)], f32x2{ taps.f2[n][0], taps.f2[n][1] }, acc);
		prevdec[s] = (__float_as_uint(acc.x) & 0xffffu) | (__float_as_uint(acc.y) << 16);
	}
	if (persist)
		__syncthreads();  // the next tile's stage 1 writes the LDS image this tile's stage 2 has just read
	}
}

// ---- FM discriminator (fm_dev, dsp_stuff.cpp:284-292) of every decimated sample against its predecessor.  The fp64
// atan2 is a pure map, so it runs here in parallel instead of inside the serial demodulator chains; TFA_2, TFA_3 and
// TX22 (tfa2.cpp:361) all consume this one array.  Only samples inside a trigger window are ever read: a piece of
// 256 samples (one wave) is computed iff a trigger lies in it or within `wmax` samples before it (wmax = the longest
// window of the registered demodulators; the first wmax samples always, a window may be open from the previous
// submit) -- about half of the benchmark workload.  Same 256 x 4 lane layout as the front end.
constexpr int kFmPersist = 4096;
__global__ __launch_bounds__(kFmThreads) void fmdev_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							      const unsigned long long *__restrict__ mask, size_t mask_stride,
							      const uint32_t *__restrict__ prevdec, int16_t *__restrict__ fmdev,
							      size_t fmdev_stride, EventBuf *__restrict__ eb, int wmax, double flag_eps,
							      int ntiles, int n_streams, int persist)
{
	// kFmPersist workgroups that take the tiles in turn, like the front end's (393 k workgroups on a low-priority stream were placed
	// only when every other queue was empty: 3.3 ms inside the batch for 0.8 ms of work; TFREC_AMD_FMDEV_PERSIST=0: a workgroup per tile)
	const int n_work = persist ? ntiles * n_streams : 1;
	for (int wi = persist ? (int)blockIdx.x : 0; wi < n_work; wi += persist ? (int)gridDim.x : 1) {
	const int s = persist ? wi / ntiles : (int)blockIdx.y, tile = persist ? wi - (wi / ntiles) * ntiles : (int)blockIdx.x;
	int tid = threadIdx.x;
	asm volatile("" : "+v"(tid));
	const int m0 = tile * kFmTile;
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	// Wave priority 1: the pass heads the TFA_2 family's biquad stream, one of the two chains that end at the period.  At
	// priority 0 the pipeline has two stable states -- this pass 1.8 ms inside the batch and the batch 6.6 ms, or 2.7 and
	// 7.3 (two runs in five) --, at priority 1 or 2 only the first (nine runs of nine).
#ifndef TFREC_AMD_FMDEV_PRIO
#define TFREC_AMD_FMDEV_PRIO 1
#endif
	__builtin_amdgcn_s_setprio(TFREC_AMD_FMDEV_PRIO);
	// A sample is read by a demodulator only inside a trigger window, i.e. if a trigger lies at most wmax - 1 samples
	// before it (the first wmax samples of a submit may belong to a window left open by the previous one).  Decided
	// per wave = per 256 samples: the mask words of [first - wmax, last], one per lane, one ballot.
	{
		const int mw = m0 + 4 * (tid & ~63);  // the wave's first sample
		if (mw >= wmax) {
			const int w0 = (mw - wmax) >> 6, w1 = (mw + 255) >> 6;
			const int w = w0 + (tid & 63);
			const bool hit = w <= w1 && mask[(size_t)s * mask_stride + w] != 0ull;
			if (__ballot(hit) == 0ull)
				continue;
		}
	}
	const uint4 v = *reinterpret_cast<const uint4 *>(drow + m0 + 4 * tid);
	const uint32_t pw = (m0 + 4 * tid) > 0 ? drow[m0 + 4 * tid - 1] : prevdec[s];
	const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
	int dv[4];
	bool unc[4];
	// The cross terms (dsp_stuff.cpp:288-289) as 32-bit integers: cr = I*pI + Q*pQ is one v_dot2_i32_i16 on the packed
	// samples as they are stored, cj = Q*pI - I*pQ two 24-bit multiplies.  |cj| < 2^31 always; cr wraps only for
	// I = Q = pI = pQ = -32768, where cj = 0: an exact direction, and those are recomputed in fp64 below.  (In fp64 from
	// the start: four conversions, four products and two sums per sample instead of seven instructions.)
	typedef short s16x2 __attribute__((ext_vector_type(2)));
	uint32_t pword = pw;
	int pI = (int)(int16_t)(pw & 0xffff), pQ = (int)pw >> 16;
#pragma unroll
	for (int o = 0; o < 4; o++) {
		const int I = (int)(int16_t)(w4[o] & 0xffff), Q = (int)w4[o] >> 16;
		const int cr = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, w4[o]), __builtin_bit_cast(s16x2, pword), 0, false);
		const int cj = Q * pI - I * pQ;  // (16-bit factors: v_mul_i32_i24 / v_mad_i32_i24)
		const double crd = (double)cr, cjd = (double)cj;
		const AtanRed red = atan2_reduce(cjd, crd, kAtanPolyFront);
		double sv;
		// The exactly representable directions (cj == 0 || cr == 0 || |cj| == |cr|: red.num == 0, see atan2_reduce) are
		// one sample in 2500 of a noise stream: ONE wave-uniform test keeps their divergent chain of cases out of the common
		// path.  (One test for a lane's four samples -- four straight-line evaluations side by side -- was tried: a tenth of
		// the waves then take the slow branch, and 74 registers instead of 28 made the pass 60 % slower inside the batch.)
		if (__builtin_expect(__ballot(red.num == 0.0) != 0ull, 0)) {
			unc[o] = fm_dev_fast(((double)I) * pI + ((double)Q) * pQ, ((double)Q) * pI - ((double)I) * pQ, &sv, kAtanPolyFront, flag_eps);
		} else {
			sv = atan2_reduced(red, cjd, crd, kAtanPolyFront);
			unc[o] = fabs(sv - rint(sv)) < flag_eps;
		}
		dv[o] = (int)sv;  // |v| <= 16384: the range tests of d2i (x86's out-of-range result) can never fire
		pword = w4[o];
		pI = I;
		pQ = Q;
	}
	// next to a truncation boundary (~2e-9 of the samples): fm_resolve_kernel decides it (one test for the lane's four samples)
	if (__builtin_expect(unc[0] || unc[1] || unc[2] || unc[3], 0)) {
#pragma unroll
		for (int o = 0; o < 4; o++)
			if (unc[o]) {
				const uint32_t i = atomicAdd(&eb->fm_pending, 1u);
				if (i < (uint32_t)kFmListCap)
					eb->fm_list[i] = ((unsigned long long)(uint32_t)s << 32) | (uint32_t)(m0 + 4 * tid + o);
			}
	}
	*reinterpret_cast<uint2 *>(fmdev + (size_t)s * fmdev_stride + m0 + 4 * tid) =
		make_uint2(((uint32_t)dv[0] & 0xffffu) | ((uint32_t)dv[1] << 16), ((uint32_t)dv[2] & 0xffffu) | ((uint32_t)dv[3] << 16));
	}
}

// ---- BASELINE config 5: 15.36 MS/s u8 IQ -> 1.536 MS/s int16 (I,Q).  The reference has no such stage; it is defined
// (oracle/tfrec_oracle.c: orc_decim10) in the reference's FIR style: 60 int16 taps (Hamming-windowed sinc, cut-off
// 768 kHz, unity DC gain), arithmetic >>16 per tap, int16 store, 10:1:
//   y[m] = int16( sum_{n<60} ( x[10 m - 50 + n] * h[n] ) >> 16 ),  x = (u8 - 128) << 6   (so (x*h)>>16 = ((u8-128)*h)>>10)
// This is the stage that streams HBM (10x the bytes per output of the standard front end) and it is 87 % of the
// configuration's front-end arithmetic: 60 packed FMAs per output.  One 128-thread workgroup per tile of 1024 outputs:
//   * the tile's 20 KiB of raw bytes + 112 B of history are staged into LDS with COALESCED 16-byte loads.  (A lane
//     reading its own 260 bytes straight from global memory -- round 4's first version -- touches every cache line with
//     eight different load instructions: the kernel was bound by L1 transactions, 2.8 ms per 256 streams; 16 outputs per
//     lane made it worse.)
//   * a lane makes kR10 = 8 consecutive outputs of both rails from the 130 raw samples they span: lane stride 40 dwords.
//     The LDS image has ONE pad dword behind every 40: the stride becomes 41 dwords and the 64 lanes of a read hit 64
//     different banks (round 3 read the unpadded image: stride 20 dwords at 4 outputs per lane, an 8-way conflict).
//   * a sample's (up to six) FMAs go to different accumulators back to back: left to itself the scheduler chains the
//     FMAs of an accumulator and pays a wait state behind two out of three (round 3: 156 s_nop for 240 FMAs).
constexpr int kR10 = 8;                 // outputs per lane
constexpr int kT10 = 1024;              // outputs per workgroup
constexpr int kThreads10 = kT10 / kR10; // 128
constexpr int kTail10 = 112;            // 56 complex samples of history (50 needed), 16-byte multiple
constexpr int kD10 = (2 * (60 + 10 * (kR10 - 1)) + 3) / 4;  // raw dwords a lane reads: 130 complex samples = 65 dwords
constexpr int kRawDw10 = (kTail10 + 20 * kT10) / 4;          // logical dwords of a tile's LDS image (dword 0 = byte 20 m0 - 112)
constexpr int kLaneDw10 = 20 * kR10 / 4;                     // 40: a lane's stride in logical dwords, = the pad interval
static_assert(kLaneDw10 % 4 == 0, "a 16-byte chunk never straddles a pad");
__device__ __forceinline__ constexpr int pad10(int d) { return d + d / kLaneDw10; }  // logical -> physical dword
__device__ __constant__ const int kTaps10[60] = {
	9,    27,   48,   72,   98,   121,  135,  132,  104,  44,   -53,  -185, -343, -511, -668,
	-783, -826, -765, -572, -230, 269,  916,  1690, 2552, 3452, 4333, 5133, 5793, 6265, 6512,
	6510, 6265, 5793, 5133, 4333, 3452, 2552, 1690, 916,  269,  -230, -572, -765, -826, -783,
	-668, -511, -343, -185, -53,  44,   104,  132,  135,  121,  98,   72,   48,   27,   9,
};

__global__ __launch_bounds__(kThreads10) void decim10_kernel(const uint8_t *__restrict__ iq, size_t stride, long n_out,
							     const uint8_t *__restrict__ tail_in, uint8_t *__restrict__ tail_out,
							     uint32_t *__restrict__ out, size_t out_stride)
{
	__shared__ uint32_t raw[pad10(kRawDw10) + 4];
	const int s = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
	const long m0 = (long)tile * kT10;
	__builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 2);  // fp32 rounding toward -inf (see frontend_kernel, stage 1)
	const long nbytes = 20L * n_out;
	const uint8_t *src = iq + (size_t)s * stride;
	// ---- stage the tile: chunk c = logical dwords 4c .. 4c + 3 = stream bytes 20 m0 - 112 + 16 c ..; the submit's first
	// tile takes its first seven chunks from the previous submit's tail
	const long base = 20L * m0 - kTail10;
	constexpr int kChunks = kRawDw10 / 4;
#pragma unroll
	for (int c0 = 0; c0 < kChunks; c0 += kThreads10) {
		const int c = c0 + tid;
		if (c0 + kThreads10 <= kChunks || c < kChunks) {
			const long bo = base + 16L * c;
			const uint4 v = bo >= 0 ? *reinterpret_cast<const uint4 *>(src + bo)
						: *reinterpret_cast<const uint4 *>(tail_in + (size_t)s * kTail10 + (kTail10 + bo));
			uint32_t *d = raw + 4 * c + c / (kLaneDw10 / 4);
			d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
		}
	}
	if (tile == (int)gridDim.x - 1 && tid < kTail10 / 16)
		*reinterpret_cast<uint4 *>(tail_out + (size_t)s * kTail10 + 16 * tid) =
			*reinterpret_cast<const uint4 *>(src + nbytes - kTail10 + 16 * tid);
	__syncthreads();
	// ---- the lane's outputs m0 + 8 tid + o: raw samples [10 m - 50, 10 m + 80) = logical dwords 40 tid + 3 + w, w < 65
	const uint32_t *lp = raw + (kLaneDw10 + 1) * tid;  // physical dword of logical 40 tid
	// (d*h) >> 10 per tap as one fp32 FMA in round-toward-minus-infinity mode, both rails per v_pk_fma_f32: see stage 1 of
	// frontend_kernel.  The 60 taps' terms sum to less than 2^14.  The raw dwords come from LDS eight at a time, one group
	// ahead of the one being consumed (all 65 up front cost 130 registers).
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	const float kMagic = 12582912.0f;  // 2^23 + 2^22
	f32x2 acc[kR10];
#pragma unroll
	for (int o = 0; o < kR10; o++)
		acc[o] = f32x2{ kMagic, kMagic };
	constexpr int kGrp = 8, kGroups10 = (kD10 + kGrp - 1) / kGrp;
	uint32_t cur[kGrp], nxt[kGrp];
#pragma unroll
	for (int k = 0; k < kGrp; k++)
		cur[k] = lp[pad10(3 + k)];
#pragma unroll
	for (int g = 0; g < kGroups10; g++) {
#pragma unroll
		for (int k = 0; k < kGrp; k++)
			nxt[k] = kGrp * (g + 1) + k < kD10 ? lp[pad10(3 + kGrp * (g + 1) + k)] : 0u;
#pragma unroll
		for (int k = 0; k < kGrp; k++) {
			const int w = kGrp * g + k;  // one dword = two complex samples
			if (w < kD10) {
				const uint32_t v = cur[k] ^ 0x80808080u;  // bytes become two's complement (u8 - 128)
				const f32x2 x[2] = { f32x2{ (float)(signed char)(v), (float)(signed char)(v >> 8) },
						     f32x2{ (float)(signed char)(v >> 16), (float)((int)v >> 24) } };
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const int c = 2 * w + h;  // sample index relative to 10 m - 50
#pragma unroll
					for (int o = 0; o < kR10; o++) {
						const int n = c - 10 * o;
						if (n >= 0 && n < 60) {
							const float hs = (float)kTaps10[n] * (1.0f / 1024.0f);
							acc[o] = __builtin_elementwise_fma(x[h], f32x2{ hs, hs }, acc[o]);
						}
					}
					__builtin_amdgcn_sched_barrier(0);  // (see above: a sample's FMAs stay together, accumulator by accumulator)
				}
			}
		}
#pragma unroll
		for (int k = 0; k < kGrp; k++)
			cur[k] = nxt[k];
	}
	uint32_t ow[kR10];
#pragma unroll
	for (int o = 0; o < kR10; o++)  // the int16 store: the low half of the accumulator's mantissa
		ow[o] = (__float_as_uint(acc[o].x) & 0xffffu) | (__float_as_uint(acc[o].y) << 16);
	uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)s * out_stride + m0 + kR10 * tid);
#pragma unroll
	for (int q = 0; q < kR10 / 4; q++)
		dst[q] = make_uint4(ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3]);
}

hipError_t launch_decim10(hipStream_t st, const uint8_t *iq, size_t stride, int n_streams, int n_blocks,
			  const uint8_t *tail_in, uint8_t *tail_out, uint32_t *out, size_t out_stride)
{
	const long n_out = (long)n_blocks * (TFREC_AMD_BLOCK_BYTES / 2);  // complex samples at 1.536 MS/s
	static_assert(kBlockDec * 4 % kT10 == 0, "decim10_kernel has no partial tiles: a block is a whole number of them");
	dim3 grid((unsigned)(n_out / kT10), n_streams);
	hipLaunchKernelGGL(decim10_kernel, grid, dim3(kThreads10), 0, st, iq, stride, n_out, tail_in, tail_out, out, out_stride);
	return hipGetLastError();
}

hipError_t launch_frontend(hipStream_t st, const uint8_t *iq, size_t stride, int n_streams, int n_blocks,
			   const uint8_t *tail_in, uint8_t *tail_out, uint32_t *dec, size_t dec_stride,
			   unsigned long long *mask, size_t mask_stride, uint32_t *prevdec, int thresh, const FrontTaps &taps,
			   bool in16)
{
	const int m_total = n_blocks * kBlockDec;
	// experiment knobs: extra dynamic LDS per workgroup (caps the front end's workgroups per CU); a fixed number of workgroups
	static const int pad = TFREC_KNOB_INT("FE_LDS_PAD", 0, 0, 64 << 10);
	// (config 5's int16 entry keeps a workgroup per tile: behind the 10:1 stage, which it waits for, the persistent form measured 4 % slower)
	static const int persist_u8 = TFREC_KNOB_INT("FE_PERSIST", kFrontPersist, 0, 1 << 20);
	const int persist = in16 ? 0 : persist_u8;
	const dim3 grid = persist ? dim3((unsigned)std::min<long>(persist, (long)(m_total / kTileDec) * n_streams)) : dim3(m_total / kTileDec, n_streams);
	if (in16)
		hipLaunchKernelGGL(frontend_kernel<true>, grid, dim3(kFrontThreads), pad, st, iq, stride, m_total, tail_in, tail_out,
				   dec, dec_stride, mask, mask_stride, prevdec, thresh, taps, n_streams, persist);
	else
		hipLaunchKernelGGL(frontend_kernel<false>, grid, dim3(kFrontThreads), pad, st, iq, stride, m_total, tail_in, tail_out,
				   dec, dec_stride, mask, mask_stride, prevdec, thresh, taps, n_streams, persist);
	return hipGetLastError();
}

// after the front end (and the auto-threshold pass, which rewrites the mask)
// Parity probe (tfrec_amd_fm_dev_probe): the device's fm_dev -- fast path, exact slow path and log -- on arbitrary
// quadruples (kind 0: int32 ar, aj, br, bj) or cross terms (kind 1: int64 cr, cj); kind 2: fm_dev_nrzs of quadruples.
__global__ __launch_bounds__(256) void fm_probe_kernel(const int32_t *__restrict__ quads, size_t n, int32_t *__restrict__ out,
							 EventBuf *__restrict__ eb, int kind)
{
	const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (k >= n)
		return;
	if (kind == 0) {
		const int4 q = reinterpret_cast<const int4 *>(quads)[k];
		out[k] = fm_dev(q.x, q.y, q.z, q.w, eb, kAtanPolyFront);
	} else if (kind == 2) {  // fm_dev_nrzs (dsp_stuff.cpp:269-279) incl. its +-1e9 clamp
		const int4 q = reinterpret_cast<const int4 *>(quads)[k];
		out[k] = fm_dev_nrzs(q.x, q.y, q.z, q.w);
	} else {
		const longlong2 c = reinterpret_cast<const longlong2 *>(quads)[k];
		out[k] = fm_dev_cross((double)c.x, (double)c.y, eb, kAtanPolyFront);
	}
}

// Parity probe (tfrec_amd_iir_probe): iir2::step (dsp_stuff.cpp:47-56) over a sequence of doubles from the zero state, one
// lane, in the two forms the kernels use -- form 0: iir_step, the reference's association as written; form 1: iir_step_t, the
// 3-multiply form of the biquad passes and of WHB stage 2 (dsp_dev.h).  Both must reproduce the reference bit for bit.
__global__ __launch_bounds__(64) void iir_probe_kernel(const double *__restrict__ in, size_t n, BiquadCoef c, double *__restrict__ out, int form)
{
	if (blockIdx.x != 0 || threadIdx.x != 0)
		return;
	Biquad f = { 0.0, 0.0, 0.0, 0.0 };
	if (form == 0) {
		for (size_t k = 0; k < n; k++)
			out[k] = iir_step(f, c, in[k]);
	} else {
		BiquadT t = iirt_enter(f, c);
		for (size_t k = 0; k < n; k++)
			out[k] = iir_step_t(f, t, c, in[k]);
	}
}

hipError_t launch_iir_probe(hipStream_t st, const double *in, size_t n, const BiquadCoef &c, double *out, int form)
{
	hipLaunchKernelGGL(iir_probe_kernel, dim3(1), dim3(64), 0, st, in, n, c, out, form);
	return hipGetLastError();
}

hipError_t launch_fm_probe(hipStream_t st, const int32_t *quads, size_t n, int32_t *out, EventBuf *eb, int kind)
{
	hipLaunchKernelGGL(fm_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, quads, n, out, eb, kind);
	return hipGetLastError();
}

// The samples fmdev_kernel flagged, decided by the exact slow path (fm_resolve.h) and logged for the host's check.
// Launched behind every fmdev_kernel; normally nothing is pending and every workgroup returns at once.
__global__ __launch_bounds__(64) void fm_resolve_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							   const uint32_t *__restrict__ prevdec, int16_t *__restrict__ fmdev,
							   size_t fmdev_stride, EventBuf *__restrict__ eb, int n_streams, int m_total,
							   double flag_eps)
{
	const uint32_t pending = eb->fm_pending;
	if (pending == 0)
		return;
	const size_t nthreads = (size_t)gridDim.x * 64, t0 = (size_t)blockIdx.x * 64 + threadIdx.x;
	auto one = [&](int s, int m, bool listed) {
		const uint32_t *drow = dec + (size_t)s * dec_stride;
		const uint32_t w = drow[m], pw = m > 0 ? drow[m - 1] : prevdec[s];
		const int I = (int)(int16_t)(w & 0xffff), Q = (int)w >> 16, pI = (int)(int16_t)(pw & 0xffff), pQ = (int)pw >> 16;
		const double cr = ((double)I) * pI + ((double)Q) * pQ, cj = ((double)Q) * pI - ((double)I) * pQ;
		double v;
		const bool unc = fm_dev_fast(cr, cj, &v, kAtanPolyFront, flag_eps);
		fmdev[(size_t)s * fmdev_stride + m] = (int16_t)((listed || unc) ? fm_dev_slow(cr, cj, v, eb) : d2i(v));
	};
	if (pending <= (uint32_t)kFmListCap) {
		for (size_t i = t0; i < pending; i += nthreads) {
			const unsigned long long e = eb->fm_list[i];
			one((int)(e >> 32), (int)(uint32_t)e, true);
		}
	} else {  // more flagged samples than the list holds (a periodic input can repeat one direction): redo the whole submit
		for (size_t i = t0; i < (size_t)n_streams * m_total; i += nthreads)
			one((int)(i / m_total), (int)(i % m_total), false);
	}
}

hipError_t launch_fmdev(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			size_t mask_stride, const uint32_t *prevdec, int16_t *fmdev, size_t fmdev_stride, EventBuf *eb,
			int n_streams, int n_blocks, int wmax, double flag_eps)
{
	const int m_total = n_blocks * kBlockDec;
	static const int fm_persist = TFREC_KNOB_INT("FMDEV_PERSIST", kFmPersist, 0, 1 << 20);
	const dim3 grid = fm_persist ? dim3((unsigned)std::min<long>(fm_persist, (long)(m_total / kFmTile) * n_streams)) : dim3(m_total / kFmTile, n_streams);
	hipLaunchKernelGGL(fmdev_kernel, grid, dim3(kFmThreads), 0, st, dec, dec_stride, mask, mask_stride, prevdec, fmdev,
			   fmdev_stride, eb, wmax, flag_eps, m_total / kFmTile, n_streams, fm_persist);
	// one-wave workgroups: the kernel normally has nothing to do, and a 256-thread workgroup waits until a CU has four
	// wave slots and their registers free at once -- up to a millisecond on the stream that sets the batch period
	hipLaunchKernelGGL(fm_resolve_kernel, dim3(512), dim3(64), 0, st, dec, dec_stride, prevdec, fmdev, fmdev_stride, eb,
			   n_streams, m_total, flag_eps);
	return hipGetLastError();
}

}  // namespace tfrec
