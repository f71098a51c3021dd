// tfrec_amd/csrc/chains2.hip -- window-parallel demodulator/decoder pipeline (the product path).
//
// Same results as the serial reference chains of chains.hip (and therefore as fsk_demod::process +
// the plugins, fm_demod.cpp:34-56, tfa1.cpp:143-190, tfa2.cpp:346-442, whb.cpp:632-707), re-cut so that
// only what the reference really serialises stays serial:
//
//   K2  windows_kernel     wave per stream: scan over the trigger mask -> per (stream, slot) the list of trigger
//                          windows (tfa1.cpp:147-149,179 / tfa2.cpp:351-355,428 / whb.cpp:636-641,691: a window opens
//                          at a pwr>thresh sample while the counter is 0 and its flush fires W-1 samples after the
//                          last trigger), virtual slot numbering, work queues.
//   K3  spec_biquad_kernel (speculate, repair, second repair) + fix_biquad_kernel (verify): the fp64 biquads
//                          (iir2::step) are the one recurrence whose state crosses windows.  TFA_2/TFA_3/TX22 read
//                          the fm_dev array of fmdev_kernel, WHB computes fm_dev_nrzs on the fly; outputs are the
//                          truncated integers the slicers consume.  Lane per 4096-sample segment (see K3 below).
//   K4  slicer_kernel      lane per WINDOW (work queue, long windows first): the bit slicers of TFA_1 and the
//                          TFA_2 family are window-local state machines (state reset at window open/close) --
//                          except tfa2's last_bit_idx, which is never reset (tfa2.cpp:325-334).  It can only
//                          influence a window through its first candidate edge, so windows are run assuming a
//                          far-away last edge and the assumption is checked (and the window re-run exactly) in K5.
//                          Output: the bits handed to decoder::store_bit, packed.
//   K4a mark_kernel, K4b coop_slicer_kernel: windows of 4096 samples and more: wave per window (see K4b below).
//   K4' whb_demod_kernel   wave per stream: WHB stage 2 (decision-level biquad, phase-change detector); the
//                          demodulator needs the decoder's has_sync() (whb.cpp:653, 677, 693), which is tracked
//                          with a lane-parallel evaluation of the (GF(2)-linear) sync search.  Output: bit runs.
//                          In its tail (K4''): whb_decoder::store_bit over the runs, lane per window, then the
//                          stream's flush events and decoder state.  <false, false>: the decision levels from a
//                          lane-parallel scan (speculated), every decision recorded;
//   K4v whb_verify_kernel  the reference's own recurrence over those samples, four streams per wave, compares the
//                          decisions and carries the exact filter state; failed streams are redone by
//                          whb_demod_kernel<true, true> (the exact form), their events retracted.
//   K5  decode_kernel      lane per window: the decoders (store_bit) over the window's packed bits;
//       commit_kernel      lane per (stream, slot): walks the windows in order: checks the tfa2 speculation (a chain
//                          with a window to re-run goes to commit_wave_kernel, wave per chain), overlays the windows'
//                          rdata bytes, emits the flush events, commits ChainState for the next submit.
// launch_pipeline (end of file) puts them on seven streams: the stages of four consecutive submits run beside each
// other.
#include <stdlib.h>

#include <algorithm>

#include "decoder_dev.h"
#include "knobs.h"
#include <type_traits>
#include "whb_chain_asm.h"

namespace tfrec {

// experiment knobs (knobs.h): read from the environment only in the -DTFREC_AMD_EXPERIMENTS build

constexpr int kSpecLbi = -(1 << 30);  // "last edge far in the past"

// The WHB test hooks (knobs.h: forced failures of the check, a perturbed frozen average) exist in the experiments build only
__device__ __forceinline__ int whb_hook_perturb(const WinTables &T)
{
#ifdef TFREC_AMD_EXPERIMENTS
	return T.whb_test_perturb;
#else
	return 0;
#endif
}
__device__ __forceinline__ int whb_hook_force_fail(const WinTables &T)
{
#ifdef TFREC_AMD_EXPERIMENTS
	return T.whb_force_fail;
#else
	return 0;
#endif
}

// tfa1.cpp:159 "mark_lvl = (int)(mark_lvl * 0.95)" for mark_lvl >= 0 (the peak detector never goes negative) without the
// trip through a double: the double nearest 0.95 is 0.95 - 4.4e-17, so the exact product is m * 19 / 20 minus less than
// 0.22 ulp of itself -- it rounds back to m * 19 / 20 where that is an integer and stays strictly inside
// (floor, floor + 1) elsewhere (the fraction is a multiple of 1 / 20): (int) of it is floor(m * 19 / 20) = m - ceil(m / 20).
__device__ __forceinline__ int tfa1_decay(int m) { return m - (int)(((uint32_t)m + 19u) / 20u); }

// Wave priority of the LATENCY-bound kernels (serial chains per lane or per wave: slicers, WHB stage 2 and its check):
// experiment knob, see profiles/NOTES.md (round 3)
#ifndef TFREC_AMD_LAT_PRIO
#define TFREC_AMD_LAT_PRIO 0
#endif
// experiment: cap the registers of the latency-bound kernels (more waves of the throughput kernels fit beside them)
#ifdef TFREC_AMD_LAT_VGPRS
#define TFREC_LAT_VGPR_ATTR __attribute__((amdgpu_num_vgpr(TFREC_AMD_LAT_VGPRS)))
#else
#define TFREC_LAT_VGPR_ATTR
#endif
__device__ __forceinline__ void latency_prio()
{
	if (TFREC_AMD_LAT_PRIO > 0)
		__builtin_amdgcn_s_setprio(TFREC_AMD_LAT_PRIO);
}

// tfrec_amd_stats counters of the cooperative slicers: counted per wave in registers and added to WinTables::stats ONCE, when the
// wave ends (an atomic per group -- 180 k per batch on two addresses -- queued up in the L2 and every wave's next wait on
// memory sat behind it: the batch 25 % longer, profiles/r06_ab_power_sum.txt).  The statistics builds (-DTFREC_AMD_COOPSTAT /
// _VECSTAT / _PROFILE_WHB) use the same slots for their own figures: there the counters are left out.
constexpr int kStatTfa1Scalar = 7, kStatTfa2Scalar = 8, kStatTfa1Vector = 9, kStatTfa2Vector = 10;
struct GroupStats {
	int scalar, vector;  // groups of 64 steps left to the scalar walk / done a step per lane (wave-uniform)
};
__device__ __forceinline__ void stat_flush(const WinTables &T, const GroupStats &g, int slot_scalar, int slot_vector)
{
#if !defined(TFREC_AMD_COOPSTAT) && !defined(TFREC_AMD_VECSTAT) && !defined(TFREC_AMD_PROFILE_WHB)
	if ((threadIdx.x & 63) == 0) {
		if (g.scalar)
			atomicAdd(&T.stats[slot_scalar], (unsigned long long)g.scalar);
#ifdef TFREC_AMD_EXPERIMENTS  // (nearly every wave has some: 29 k more atomics per launch -- counted where a test asks for them)
		if (g.vector)
			atomicAdd(&T.stats[slot_vector], (unsigned long long)g.vector);
#endif
	}
#endif
}

// demodulator::start (decoder.cpp:118-122) applied once per block boundary between two blocks
__device__ __forceinline__ int rebase_lbi(int lbi, int from_block, int to_block)
{
	return lbi ? lbi - kIndexSpan * (to_block - from_block) : 0;
}

// Packed bit output of a slicer lane.  A completed 32-bit word is parked in a register and written by
// chunk_end(), which the caller invokes unconditionally once per 32-sample chunk (a slicer emits < 0.5 bit per
// sample, so at most one word completes per chunk): the hot loop has no conditional global store, and the
// prefetch waits never have to cover a store issued a moment ago.
struct BitWriter {
	uint32_t *base;
	uint32_t acc;
	int n;
	uint32_t pend;
	int pend_idx;  // -1: nothing parked
	__device__ __forceinline__ void put(int bit)
	{
		acc |= (uint32_t)bit << (n & 31);
		n++;
		if ((n & 31) == 0) {
			if (pend_idx >= 0)
				base[pend_idx] = pend;  // second completion inside one chunk (only the 16 trailing bits can do it)
			pend = acc;
			pend_idx = (n >> 5) - 1;
			acc = 0;
		}
	}
	// cnt in [1, 31] bits at once, bit k of v = the k-th of them (the run of an accepted edge: tfa2.cpp:399-404)
	__device__ __forceinline__ void put_bits(uint32_t v, int cnt)
	{
		const int sh = n & 31;
		const unsigned long long a = (unsigned long long)acc | ((unsigned long long)v << sh);
		n += cnt;
		if (sh + cnt >= 32) {
			if (pend_idx >= 0)
				base[pend_idx] = pend;
			pend = (uint32_t)a;
			pend_idx = (n >> 5) - 1;
			acc = (uint32_t)(a >> 32);
		} else {
			acc = (uint32_t)a;
		}
	}
	__device__ __forceinline__ void chunk_end()
	{
		const int idx = pend_idx >= 0 ? pend_idx : (n >> 5);
		base[idx] = pend_idx >= 0 ? pend : acc;  // rewriting the partial word is harmless
		pend_idx = -1;
	}
	__device__ __forceinline__ void finish()
	{
		chunk_end();
		if (n & 31)
			base[n >> 5] = acc;
	}
};

// ------------------------------------------------------------------------------------------------ K1b auto threshold
// fsk_demod::process in auto mode (thresh_mode == 1, fm_demod.cpp:58-73): per block of 8192 decimated samples
//   triggered     = samples at which at least one demodulator is inside its window
//   triggered_avg = (31*triggered_avg + triggered)/32;   every 4th block: avg >= len/32 -> thresh += 2,
//                   avg <= len/64 && thresh > 50 -> thresh -= 2          (len = 16384)
// The threshold of block b+1 depends on block b, so a stream is scanned block by block; all demodulators use
// the same trigger test, so "some demodulator is in its window" = "within Wmax samples after a trigger" with Wmax
// the largest window of the registered demodulators.  One wave per stream: 64 samples per step (coalesced),
// ballot -> mask word, wave-uniform bookkeeping.  Rewrites the trigger mask the front end produced.
__global__ __launch_bounds__(64) void threshold_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						       unsigned long long *__restrict__ mask, size_t mask_stride, int n_blocks,
						       FskState *__restrict__ fsk, int wmax)
{
	const int s = blockIdx.x;
	const int lane = threadIdx.x;
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	unsigned long long *mrow = mask + (size_t)s * mask_stride;
	FskState st = fsk[s];
	int last_trig = st.last_trig;  // relative to sample 0 of this submit (very negative: none)
	for (int b = 0; b < n_blocks; b++) {
		int triggered = 0;
		st.runs++;
		for (int w = b * (kBlockDec / 64); w < (b + 1) * (kBlockDec / 64); w++) {
			const uint32_t cw = drow[(w << 6) + lane];
			const int I = (int)(int16_t)(cw & 0xffff), Q = (int)cw >> 16;
			const unsigned long long m = __ballot((abs(I) + abs(Q)) > st.thresh);
			if (lane == 0)
				mrow[w] = m;
			// samples of this word that lie within wmax after the last trigger (windows are >= 355 > 64 long:
			// everything after the word's first trigger is inside)
			const int g0 = w << 6;
			const int first = m ? __builtin_ctzll(m) : 64;
			int carried = last_trig + wmax - g0;  // samples from g0 on still covered by the earlier trigger
			carried = carried < 0 ? 0 : (carried > first ? first : carried);
			triggered += carried + (64 - first);
			if (m)
				last_trig = g0 + 63 - __builtin_clzll(m);
		}
		st.triggered_avg = (31 * st.triggered_avg + triggered) / 32;
		if ((st.runs & 3) == 0) {
			if (st.triggered_avg >= kIndexSpan / 32)
				st.thresh += 2;
			else if (st.triggered_avg <= kIndexSpan / 64 && st.thresh > 50)
				st.thresh -= 2;
		}
	}
	if (lane == 0) {
		const int M = n_blocks * kBlockDec;
		st.last_trig = last_trig - M < -(1 << 28) ? -(1 << 28) : last_trig - M;
		fsk[s] = st;
	}
}

hipError_t launch_threshold(hipStream_t st, const uint32_t *dec, size_t dec_stride, unsigned long long *mask,
			    size_t mask_stride, int n_streams, int n_blocks, FskState *fsk, int wmax)
{
	hipLaunchKernelGGL(threshold_kernel, dim3(n_streams), dim3(64), 0, st, dec, dec_stride, mask, mask_stride, n_blocks, fsk,
			   wmax);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K2
// One WAVE per stream: 64 mask words are loaded coalesced per step, a ballot finds the non-zero ones, and
// a wave-uniform scalar walk over runs of non-zero words maintains, for every active slot of the stream, the
// window state (a window opens at a trigger sample while the timeout counter is 0 and its flush fires W-1
// samples after the last trigger: tfa1.cpp:147-149,179 / tfa2.cpp:351-355,428 / whb.cpp:636-641,691).
// A gap that closes a window is >= W-1 >= 355 samples, so it always spans whole 64-bit words: only the first
// trigger of a run's first word and the last trigger of its last word matter.
__global__ __launch_bounds__(64) void windows_kernel(const unsigned long long *__restrict__ mask, size_t mask_stride,
						     int n_streams, int n_blocks, ChainLaunch L, WinTables T, int long_window)
{
	// The small kernels between the big passes (scan, verify, decode, commit: a few hundred waves of table work) issue
	// ahead of the throughput kernels they share a SIMD with: they cost those nothing measurable and every one of them
	// stands in a stream's chain (-1 % per batch).
	__builtin_amdgcn_s_setprio(3);
	const int s = blockIdx.x;
	const int lane = threadIdx.x;
	const int M = n_blocks * kBlockDec;
	const int nwords = M >> 6;
	const unsigned long long *mrow = mask + (size_t)s * mask_stride;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	// per active slot, wave-uniform
	int W[kNSlots], t0[kNSlots], open_g[kNSlots], last_trig[kNSlots], count[kNSlots], vs[kNSlots];
	bool open[kNSlots], overflow = false;
#pragma unroll
	for (int a = 0; a < kNSlots; a++) {
		const bool act = a < L.n_active;
		W[a] = act ? L.params[a].window : 400;
		t0[a] = act ? T.timeout_carry[a * n_streams + s] : 0;
		open[a] = t0[a] > 0;
		open_g[a] = 0;
		last_trig[a] = open[a] ? t0[a] - W[a] : -(1 << 29);  // virtual trigger leaving t0 samples of window
		count[a] = 0;
		vs[a] = 0;
	}
	// work items are collected per wave in LDS and handed to the global queues with ONE atomic per queue and
	// flush (thousands of waves pushing single items contend on a handful of counters otherwise)
	constexpr int kLocal = 256;
	__shared__ uint2 litems[kNQueues][kLocal];
	int lcount[kNQueues];
#pragma unroll
	for (int q = 0; q < kNQueues; q++)
		lcount[q] = 0;
	auto flush_one = [&](int q, int &nq) {  // wave-uniform, q is a compile-time constant at every call site
		if (nq == 0)
			return;
		uint32_t base0 = 0;
		if (lane == 0)
			base0 = atomicAdd(&T.queue[q].count, (uint32_t)nq);
		base0 = __builtin_amdgcn_readfirstlane(base0);
		for (int k = lane; k < nq; k += 64)
			T.items[(size_t)q * total + base0 + k] = litems[q][k];
		nq = 0;
	};
	auto push = [&](int q, uint2 it) {  // wave-uniform; static indexing keeps lcount[] in registers
#pragma unroll
		for (int qq = 0; qq < kNQueues; qq++)
			if (qq == q) {
				if (lcount[qq] == kLocal)
					flush_one(qq, lcount[qq]);
				if (lane == 0)
					litems[qq][lcount[qq]] = it;
				lcount[qq]++;
			}
	};
	auto emit = [&](int a, int og, int close) {
		const int c = a * n_streams + s;
		if (count[a] < T.cap) {
			if (lane == 0) {
				T.open[(size_t)c * T.cap + count[a]] = og;
				T.close[(size_t)c * T.cap + count[a]] = close;
			}
			const int kind = L.params[a].kind;
			const int last = close < M ? close : M - 1;
			const int n = last - og + 1;
			if (kind < 2)  // slicer work item: queues 2*kind + {0 long, 1 short}
				push(2 * kind + (n >= long_window ? 0 : 1), make_uint2((uint32_t)c, (uint32_t)count[a]));
			if (kind == 0 && n >= long_window)  // peak-detector pieces of a long TFA_1 window
				for (int pc = 0; pc * kMarkSlots * 32 < n; pc++)
					push(7, make_uint2((uint32_t)c, (uint32_t)count[a] | ((uint32_t)pc << 17)));
			if (kind > 0) {  // the chain owns a biquad: an item per segment that starts in this window; queue 4: TFA_2
				         // family, 6: WHB
				const int nch = (n + 31) >> 5;
				const int v0 = vs[a];
				for (int v = (v0 + kSegSlots - 1) / kSegSlots * kSegSlots; v < v0 + nch; v += kSegSlots) {
					const int k = v / kSegSlots;
					if (lane == 0)
						T.segstart[(size_t)c * T.segcap + k] = make_uint2((uint32_t)count[a], (uint32_t)(v - v0));
					push(2 + 2 * kind, make_uint2((uint32_t)c, (uint32_t)k));
				}
				vs[a] = v0 + nch;
			}
		} else
			overflow = true;
	};
	for (int w0 = 0; w0 < nwords; w0 += 64) {
		const int w = w0 + lane;
		const unsigned long long m = w < nwords ? mrow[w] : 0ull;
		unsigned long long nz = __ballot(m != 0);
		const int first_bit = m ? __builtin_ctzll(m) : 0;
		const int last_bit = m ? 63 - __builtin_clzll(m) : 0;
		while (nz) {
			const int l = __builtin_ctzll(nz);
			const unsigned long long run = ~(nz >> l);  // its lowest set bit marks where the run of ones from l ends
			const int len = run ? __builtin_ctzll(run) : 64 - l;
			const int l2 = l + len - 1;
			nz = (l2 >= 63) ? 0ull : (nz & (~0ull << (l2 + 1)));
			const int first = ((w0 + l) << 6) + __builtin_amdgcn_readlane(first_bit, l);
			const int lastt = ((w0 + l2) << 6) + __builtin_amdgcn_readlane(last_bit, l2);
#pragma unroll
			for (int a = 0; a < kNSlots; a++) {
				if (a < L.n_active) {
					if (open[a] && first > last_trig[a] + W[a] - 1) {
						emit(a, open_g[a], last_trig[a] + W[a] - 1);
						count[a]++;
						open[a] = false;
					}
					if (!open[a]) {
						open[a] = true;
						open_g[a] = first;
					}
					last_trig[a] = lastt;
				}
			}
		}
	}
#pragma unroll
	for (int a = 0; a < kNSlots; a++) {
		if (a < L.n_active) {
			int tnext = 0;
			if (open[a]) {
				const int close = last_trig[a] + W[a] - 1;
				emit(a, open_g[a], close);
				count[a]++;
				if (close >= M)
					tnext = close - (M - 1);
			}
			if (lane == 0) {
				const int c = a * n_streams + s;
				T.count[c] = count[a] < T.cap ? count[a] : T.cap;
				T.cont[c] = t0[a] > 0 ? 1 : 0;
				T.timeout_next[c] = tnext;
				T.timeout_carry[c] = tnext;
				T.vtotal[c] = vs[a];
			}
		}
	}
#pragma unroll
	for (int q = 0; q < kNQueues; q++)
		flush_one(q, lcount[q]);
	if (overflow && lane == 0)
		*T.overflow = 1;
}

// ------------------------------------------------------------------------------------------------ chunk iterator
// A serial lane walks the in-window samples of ITS chain in aligned 32-sample chunks.  All lanes of a wave
// share one instruction stream (one chunk per iteration, per-sample predication), each at its own position.
constexpr int kChunk = 32;

struct ChunkDesc {
	int cb;       // first sample of the aligned chunk
	int lo, hi;   // samples [lo, hi] of the chunk belong to the window
	int j;        // window ordinal
	int flags;    // 1: lo is the window's first sample, 2: hi is the window's last sample, 4: that window closes (flush)
};

struct ChunkIter {
	const int32_t *wopen, *wclose;
	int count, M;
	int j, g, last, closed;
	__device__ __forceinline__ void init(const WinTables &T, int c, int M_)
	{
		wopen = T.open + (size_t)c * T.cap;
		wclose = T.close + (size_t)c * T.cap;
		count = T.count[c];
		M = M_;
		j = -1;
		g = 1;
		last = 0;
		closed = 0;
	}
	__device__ __forceinline__ bool next(ChunkDesc &d)
	{
		int fl = 0;
		if (g > last) {
			if (++j >= count)
				return false;
			g = wopen[j];
			const int cl = wclose[j];
			closed = cl < M;
			last = closed ? cl : M - 1;
			fl = 1;
		}
		d.cb = g & ~(kChunk - 1);
		d.lo = g;
		d.hi = d.cb + kChunk - 1 < last ? d.cb + kChunk - 1 : last;
		d.j = j;
		if (d.hi == last)
			fl |= 2 | (closed ? 4 : 0);
		d.flags = fl;
		g = d.cb + kChunk;
		return true;
	}
};

// ------------------------------------------------------------------------------------------------ K3
// The fp64 biquads (iir2::step) are the one recurrence whose state crosses windows.  They are strongly
// contracting (pole radius 0.87-0.95): a run started from the WRONG state becomes bit-identical to the true
// trajectory after a few hundred samples, and once the full state (yn, yn1 + the two last inputs) matches
// bit for bit it matches forever.  The in-window slots of a chain, numbered consecutively across windows, are
// cut into segments of kSegSlots slots (>= 3700 samples), and
//   K3a spec_biquad_kernel   lane per SEGMENT (work queue): runs the segment from a zero state (every segment, the
//                            chain's first too: the pass needs nothing of the submit before and runs on a stream of
//                            its own, PipeCtl::ks), stores the truncated outputs the slicers consume, a (yn, yn1)
//                            checkpoint per slot and the full end state;
//   K3b repair_biquad_kernel lane per SEGMENT: runs the head of the segment again, now from the END state of
//                            the previous segment's speculative run (the chain's first segment: from the true
//                            carried state), rewriting the outputs until its state
//                            equals the speculative checkpoint bit for bit -- from there on the stored outputs
//                            are the continuation of THIS run;
//   K3c fix_biquad_kernel    lane per CHAIN: walks the segments in order with the true state f.  If f equals the
//                            state K3b started segment k from (bit for bit), K3b's result for k is the true
//                            trajectory and f advances by a table look-up; otherwise (the previous segment had
//                            not converged: practically only a chain's short last segment, which has no
//                            successor) the segment is repaired serially from f.  Exactness never depends on
//                            convergence; only speed does.
// Outputs are window-relative: window j of a chain owns the 32-sample slots (open>>5)+j ... so every chunk is
// full except a window's tail, and tails may be stored whole.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load, dword aligned

__device__ __forceinline__ int win_slot0(int og, int j) { return (og >> 5) + j; }

// iir2::step without the range check of d2i: |y| <= 1.0911 * max|x| (L1 norm of the impulse responses), so the
// tfa2 outputs stay below 17 877 and the WHB stage-1 outputs below 1.3e9: v_cvt_i32_f64 truncates exactly.
__device__ __forceinline__ int iir_step_i(Biquad &f, const BiquadCoef &c, int x) { return (int)iir_step(f, c, (double)x); }

template <bool WHB>
struct K3Chunk {
	uint32_t w[WHB ? 32 : 17];
	uint32_t prevw;  // WHB: decimated sample before the chunk
};

template <bool WHB>
__device__ __forceinline__ void k3_load(K3Chunk<WHB> &ch, const void *row, int g0, uint32_t prev0)
{
	if (WHB) {
		const uint32_t *drow = static_cast<const uint32_t *>(row);
		const u32x4_a4 *p = reinterpret_cast<const u32x4_a4 *>(drow + g0);
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const u32x4_a4 v = p[i];
			ch.w[4 * i] = v.x; ch.w[4 * i + 1] = v.y; ch.w[4 * i + 2] = v.z; ch.w[4 * i + 3] = v.w;
		}
		ch.prevw = g0 > 0 ? drow[g0 - 1] : prev0;
	} else {
		const int16_t *in = static_cast<const int16_t *>(row);
		const uint32_t *base = reinterpret_cast<const uint32_t *>(in + (g0 & ~1));
		const u32x4_a4 *p = reinterpret_cast<const u32x4_a4 *>(base);
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const u32x4_a4 v = p[i];
			ch.w[4 * i] = v.x; ch.w[4 * i + 1] = v.y; ch.w[4 * i + 2] = v.z; ch.w[4 * i + 3] = v.w;
		}
		ch.w[16] = base[16];
		if (g0 & 1) {  // odd start: shift the 17 dwords down by one int16
#pragma unroll
			for (int i = 0; i < 16; i++)
				ch.w[i] = (ch.w[i] >> 16) | (ch.w[i + 1] << 16);
		}
	}
}

// Filter `nvalid` samples of a chunk (groups of 8: unpredicated while the whole group is valid).  emit(grp, g): the
// group's outputs, packed as they are stored (4 dwords of int16 pairs, WHB: 8 int32), zeros beyond nvalid.
template <bool WHB, class Emit>
__device__ __forceinline__ void k3_filter(Biquad &f, const BiquadCoef &cf, const K3Chunk<WHB> &ch, int nvalid, Emit emit)
{
	int pI = (int)(int16_t)(ch.prevw & 0xffff), pQ = (int)ch.prevw >> 16;
	BiquadT bt = iirt_enter(f, cf);
#pragma unroll
	for (int grp = 0; grp < 4; grp++) {
		uint32_t g[WHB ? 8 : 4];
#pragma unroll
		for (int i = 0; i < (WHB ? 8 : 4); i++)
			g[i] = 0;
		auto one = [&](int k) {
			int y;
			if (WHB) {
				const int I = (int)(int16_t)(ch.w[k] & 0xffff), Q = (int)ch.w[k] >> 16;
				y = (int)iir_step_t(f, bt, cf, (double)fm_dev_nrzs(I, Q, pI, pQ));  // whb.cpp:651-652
				pI = I;
				pQ = Q;
				g[k & 7] = (uint32_t)y;
			} else {
				const int x = (int)(int16_t)((ch.w[k >> 1] >> (16 * (k & 1))) & 0xffff);
				y = (int)iir_step_t(f, bt, cf, (double)x);  // tfa2.cpp:362
				g[(k & 7) >> 1] |= ((uint32_t)y & 0xffffu) << (16 * (k & 1));
			}
		};
		__builtin_amdgcn_sched_barrier(0);  // bound the live range of the per-sample products to one group
		if (nvalid >= 8 * (grp + 1)) {
#pragma unroll
			for (int k = 8 * grp; k < 8 * grp + 8; k++)
				one(k);
		} else if (nvalid > 8 * grp) {
#pragma unroll
			for (int k = 8 * grp; k < 8 * grp + 8; k++)
				if (k < nvalid)
					one(k);
		}
		emit(grp, g);
	}
}
// ... into a register image of the slot (the serial repair of fix_biquad_kernel)
template <bool WHB>
__device__ __forceinline__ void k3_filter(Biquad &f, const BiquadCoef &cf, const K3Chunk<WHB> &ch, int nvalid,
					  uint32_t (&ow)[WHB ? 32 : 16])
{
	k3_filter<WHB>(f, cf, ch, nvalid, [&](int grp, const uint32_t (&g)[WHB ? 8 : 4]) {
#pragma unroll
		for (int i = 0; i < (WHB ? 8 : 4); i++)
			ow[(WHB ? 8 : 4) * grp + i] = g[i];
	});
}

template <bool WHB>
__device__ __forceinline__ void k3_store(void *outrow, int slot, const uint32_t (&ow)[WHB ? 32 : 16])
{
	uint4 *o = reinterpret_cast<uint4 *>(static_cast<uint32_t *>(outrow) + (size_t)slot * (WHB ? 32 : 16));
#pragma unroll
	for (int i = 0; i < (WHB ? 8 : 4); i++)
		o[i] = make_uint4(ow[4 * i], ow[4 * i + 1], ow[4 * i + 2], ow[4 * i + 3]);
}

// The same store for a whole wave, transposed through LDS.  Every lane holds one slot's outputs (64 / 128 contiguous
// bytes) for a row of its own; stored lane by lane, each instruction touches 64 cache lines with 16 bytes each --
// measured (profiles/ubench/hbm_mix): 1.1 TB/s for such a kernel alone instead of 4.4-4.7, 2.76x its bytes at the
// memory side, and a coalesced reader running beside it drops to 0.47 TB/s instead of 1.1.  Through the tile each store
// instruction writes 16 rows x 64 (8 rows x 128) contiguous bytes.  ALL 64 lanes of the single-wave workgroup call this
// together; dst == nullptr: nothing to store for this lane.  Rows are padded by 16 bytes: the b128 writes are
// conflict-free, the reads 2-way.  No barrier: the LDS executes one wave's instructions in order, and a workgroup
// barrier's fence would wait for the slot loads in flight (that alone cost 25 % of these passes).
template <bool WHB>
struct K3Tile {
	static constexpr int kBytes = WHB ? 128 : 64, kStride = kBytes + 16, kSize = 64 * kStride + 64 * 8;
};
template <bool WHB>
__device__ __forceinline__ uint8_t *k3_tile_row(uint8_t *tile) { return tile + (threadIdx.x & 63) * K3Tile<WHB>::kStride; }
// (the filter has written this lane's row: k3_tile_row)
template <bool WHB>
__device__ __forceinline__ void k3_store_t(uint8_t *tile, void *dst)
{
	constexpr int RS = K3Tile<WHB>::kStride, PIECES = K3Tile<WHB>::kBytes / 16, RPI = 64 / PIECES;
	const int ln = threadIdx.x & 63;
	if (__ballot(dst != nullptr) == 0ull)
		return;
	reinterpret_cast<unsigned long long *>(tile + 64 * RS)[ln] = (unsigned long long)(uintptr_t)dst;
	__builtin_amdgcn_wave_barrier();
	const int piece = ln % PIECES, rsub = ln / PIECES;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(1))) u32x4 global_u32x4;  // (a global, not a flat store)
#pragma unroll
	for (int k = 0; k < PIECES; k++) {
		const int r = k * RPI + rsub;
		const u32x4 v = *reinterpret_cast<const u32x4 *>(tile + r * RS + 16 * piece);
		const unsigned long long a = reinterpret_cast<const unsigned long long *>(tile + 64 * RS)[r];
		if (a)
			*(global_u32x4 *)(uintptr_t)(a + 16 * piece) = v;
	}
	__builtin_amdgcn_wave_barrier();
}

struct SegWin {
	int og, n, nch, slot0;
};
__device__ __forceinline__ SegWin seg_win(const WinTables &T, int c, int j, int M)
{
	SegWin w;
	w.og = T.open[(size_t)c * T.cap + j];
	const int close = T.close[(size_t)c * T.cap + j];
	w.n = (close < M ? close : M - 1) - w.og + 1;
	w.nch = (w.n + 31) >> 5;
	w.slot0 = win_slot0(w.og, j);
	return w;
}

__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// Checkpoints of the speculative pass are kept for every kCkEvery-th slot (by slot number, so every pass agrees on
// which): a repair run can only join the speculative trajectory there, up to kCkEvery - 1 slots later than with a
// checkpoint per slot (~35 slots per run on average), for a quarter of the 16-byte lane-per-row checkpoint stores and
// loads -- partial-line accesses, the expensive kind (profiles/NOTES.md, round 2).
#ifndef TFREC_AMD_CK_EVERY
#define TFREC_AMD_CK_EVERY 4
#endif
constexpr int kCkEvery = TFREC_AMD_CK_EVERY;
static_assert((kCkEvery & (kCkEvery - 1)) == 0, "a power of two");
__device__ __forceinline__ bool ck_slot(int slot) { return (slot & (kCkEvery - 1)) == kCkEvery - 1; }

// Run the biquad over `nslots` consecutive in-window slots of chain c, starting at slot i of window j (the run
// hops to the following windows as they end).  REPAIR = false: speculative run, stores outputs and checkpoints.
// REPAIR = true: stores outputs and stops after the first slot (>= min_slots slots, >= 2 samples in) whose end
// state equals the stored checkpoint bit for bit (the two last inputs are then shared too: from there on the
// stored trajectory is the continuation of this run).  Returns the slots processed.
// position in a chain's sequence of in-window slots
struct SegCursor {
	int j, i;
	SegWin w;
};
__device__ __forceinline__ void seg_advance(SegCursor &p, const WinTables &T, int c, int M, int count)
{
	if (++p.i >= p.w.nch) {
		p.i = 0;
		p.j = p.j + 1 < count ? p.j + 1 : p.j;  // (never used past the chain's last slot)
		p.w = seg_win(T, c, p.j, M);
	}
}

__device__ __forceinline__ Biquad biquad_of(const BiquadEnd &e)
{
	Biquad f;
	f.dn1 = e.dn1; f.dn2 = e.dn2; f.yn = e.yn; f.yn1 = e.yn1;
	return f;
}
__device__ __forceinline__ BiquadEnd end_of(const Biquad &f)
{
	BiquadEnd e;
	e.dn1 = f.dn1; e.dn2 = f.dn2; e.yn = f.yn; e.yn1 = f.yn1;
	return e;
}

// K3a (MODE 0), K3b (MODE 1) and K3b' (MODE 2: segments whose predecessor's repair run did not converge are run once
// more, from THAT run's end state); WHB: the WHB chains (int32 outputs from the decimated samples) or the TFA_2-family
// chains (int16 outputs from the fm_dev array).
//
// A flat loop: per iteration every busy lane filters ONE slot of its segment, and all 64 lanes store the wave's slots
// together (k3_store_t).  A lane that finishes its segment takes the next one from the work queue by itself -- the lanes
// of a wave do not wait for each other's segments.  (They did: a repair run takes ~20 slots for most segments but the
// whole segment, 116 slots, for the 1 % whose two trajectories never become bit-identical; with one such lane in
// every second wave the repair passes took as long as the speculative pass.)  Taking a segment costs a few dependent
// table reads during which the wave stalls, so idle lanes wait until a quarter of the wave is idle (or nothing runs).
// Two slot buffers per lane: slot k (A) is filtered while slot k+1 (B) is in flight; then B moves to A and slot k+2 is
// requested.  (A third buffer -- two slots in flight throughout -- made the pass faster alone and the batch slower: 30-50
// more registers per lane, profiles/NOTES.md round 2.)
template <bool WHB, int MODE>
__global__ __launch_bounds__(64) void spec_biquad_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							 const int16_t *__restrict__ fmdev, size_t fmdev_stride, int n_streams,
							 int n_blocks, ChainLaunch L, WinTables T, int16_t *__restrict__ ld16,
							 int32_t *__restrict__ dev32, int lanes)
{
#ifdef TFREC_AMD_SPEC_PRIO
	__builtin_amdgcn_s_setprio(TFREC_AMD_SPEC_PRIO);
#endif
	constexpr bool REPAIR = MODE != 0;
#ifdef TFREC_AMD_SPEC_CLAIM  // (sensitivity experiment, see slicer_kernel)
	asm volatile("" ::: TFREC_AMD_SPEC_CLAIM);
#endif
	extern __shared__ __attribute__((aligned(16))) uint8_t k3_tile[];  // K3Tile<WHB>::kSize bytes
	const int M = n_blocks * kBlockDec;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	constexpr int q = 4 + 2 * (WHB ? 1 : 0);
	const uint32_t qcount = T.queue[q].count;
	uint32_t *head = MODE == 0 ? &T.queue[q].head : (MODE == 1 ? &T.queue[q].head2 : &T.queue[q].head3);
	const bool worker = (int)threadIdx.x < lanes;  // the other lanes only help to store
	bool busy = false, dry = !worker;
	unsigned long long stat_slots = 0ull;  // tfrec_amd_stats::biquad_repair_slots of this lane's segments (added once, at the end)
	// the lane's segment
	int c = 0, count = 0, nslots = 0, min_slots = 0, done = 0, nsamples = 0, loaded = 0;
	size_t sk = 0;
	const void *in = nullptr;
	void *out = nullptr;
	double2 *ckrow = nullptr;
	uint32_t prev0 = 0;
	BiquadCoef cf = {};
	Biquad f;
	f.dn1 = f.dn2 = f.yn = f.yn1 = 0.0;
	K3Chunk<WHB> A, B;
	double2 ckA = make_double2(0, 0), ckB = ckA;
	SegCursor pp, pl;  // processing / loading position
	pp.j = pp.i = 0;
	pp.w = SegWin{ 0, 0, 1, 0 };
	pl = pp;
	auto fetch = [&](K3Chunk<WHB> &buf, double2 &ck) {
		if (loaded < nslots) {
			k3_load<WHB>(buf, in, pl.w.og + kChunk * pl.i, prev0);
			if (REPAIR && ck_slot(pl.w.slot0 + pl.i))
				ck = ckrow[pl.w.slot0 + pl.i];
			loaded++;
			if (loaded < nslots)
				seg_advance(pl, T, c, M, count);
		}
	};
	while (true) {
		// ---- take segments
		const unsigned long long idle = __ballot(!busy && !dry), running = __ballot(busy);
		if (idle != 0ull && (running == 0ull || __builtin_popcountll(idle) >= 16)) {
			while (!busy && !dry) {  // (a segment with nothing to run is finished on the spot)
				const uint32_t idx = atomicAdd(head, 1u);
				if (idx >= qcount) {
					dry = true;
					break;
				}
				const uint2 it = T.items[(size_t)q * total + idx];
				c = (int)it.x;
				const int k = (int)it.y;
				const int a = c / n_streams, s = c - a * n_streams;
				sk = (size_t)c * T.segcap + k;
				bool run = true;
				f.dn1 = f.dn2 = f.yn = f.yn1 = 0.0;
				min_slots = 0;
				if (MODE == 0) {
					// EVERY segment from a zero state, the chain's first one too (until round 5 it started from the carried state,
					// which the chain walk of the submit before writes: the pass of submit k + 1 then had to wait for it and
					// sat on the TFA_2 family's serial stage loop; now it needs the discriminator pass and the window scan only)
				} else if (MODE == 1) {
					// segment 0 from the TRUE carried state (this pass runs behind the chain walk of the submit before),
					// segment k > 0 from the speculative end of k - 1
					f = k > 0 ? biquad_of(T.segend1[sk - 1]) : L.states[a][s].iir;
				} else {
					// the run K3b made for segment k started from the speculative end of k-1; if K3b's own run of k-1 was
					// the true one, its end state segend2[k-1] is where segment k really starts
					run = k > 0 && !(T.segfix[sk - 1] & kSegConverged);
					if (run) {
						f = biquad_of(T.segend2[sk - 1]);
						min_slots = T.segfix[sk] & ~kSegConverged;
					} else {
						T.segfix2[sk] = 0;
					}
				}
				if (!run)
					continue;
				const int left = T.vtotal[c] - k * kSegSlots;
				nslots = left < kSegSlots ? left : kSegSlots;
				if (nslots <= 0) {  // (cannot happen: the queue holds existing segments)
					if (MODE == 0)
						T.segend1[sk] = end_of(f);
					else if (MODE == 1)
						T.segfix[sk] = 0;
					else
						T.segfix2[sk] = kSegRan;
					continue;
				}
				const uint2 start = T.segstart[sk];
				count = T.count[c];
				cf = L.params[a].iir;
				in = WHB ? (const void *)(dec + (size_t)s * dec_stride) : (const void *)(fmdev + (size_t)s * fmdev_stride);
				out = WHB ? (void *)(dev32 + (size_t)s * T.slots * 32) : (void *)(ld16 + (size_t)(c - T.ld_c0) * T.slots * 32);
				ckrow = T.ckpt + (size_t)(c - T.ck_c0) * T.slots;
				prev0 = T.prevdec[s];  // not the chain state's prev_i/q: stage B of the previous submit may still be running
				pp.j = (int)start.x;
				pp.i = (int)start.y;
				pp.w = seg_win(T, c, pp.j, M);
				pl = pp;
				done = nsamples = loaded = 0;
				fetch(A, ckA);
				fetch(B, ckB);
				busy = true;
			}
		}
		if (__ballot(busy) == 0ull) {
			if (__ballot(!dry) == 0ull)
				break;
			continue;
		}
		// ---- one slot
		void *dst = nullptr;
		if (busy) {
			const int nv = pp.w.n - kChunk * pp.i < kChunk ? pp.w.n - kChunk * pp.i : kChunk;
			uint4 *row = reinterpret_cast<uint4 *>(k3_tile_row<WHB>(k3_tile));  // (the transposed reads of the last slot were issued before)
			k3_filter<WHB>(f, cf, A, nv, [&](int grp, const uint32_t (&g)[WHB ? 8 : 4]) {
				if (WHB) {
					row[2 * grp] = make_uint4(g[0], g[1], g[2], g[3]);
					row[2 * grp + 1] = make_uint4(g[WHB ? 4 : 0], g[WHB ? 5 : 0], g[WHB ? 6 : 0], g[WHB ? 7 : 0]);
				} else {
					row[grp] = make_uint4(g[0], g[1], g[2], g[3]);
				}
			});
			dst = static_cast<uint32_t *>(out) + (size_t)(pp.w.slot0 + pp.i) * (WHB ? 32 : 16);
			nsamples += nv;
			done++;
			bool conv = false;
			const bool at_ck = ck_slot(pp.w.slot0 + pp.i);
			if (!REPAIR) {
				if (at_ck)
					ckrow[pp.w.slot0 + pp.i] = make_double2(f.yn, f.yn1);
			} else {  // the state equals the speculative checkpoint bit for bit (the two last inputs are then shared too):
				  // from here on the stored trajectory is the continuation of this run
				conv = at_ck && same_bits(f.yn, ckA.x) && same_bits(f.yn1, ckA.y) && nsamples >= 2 && done >= min_slots;
			}
			if (conv || done >= nslots) {
				busy = false;
				if (MODE == 0) {
					T.segend1[sk] = end_of(f);
				} else if (MODE == 1) {
					T.segfix[sk] = done | (conv ? kSegConverged : 0);
					stat_slots += (unsigned long long)done;
					if (!conv) {
						T.segend2[sk] = end_of(f);
						atomicAdd(&T.stats[1], 1ull);
					}
				} else {
					T.segfix2[sk] = done | (conv ? kSegConverged : 0) | kSegRan;
					if (!conv)
						T.segend3[sk] = end_of(f);
				}
			} else {
				seg_advance(pp, T, c, M, count);
			}
		}
		// (for every lane, busy or not: as part of the branch above the buffers -- loop-carried in both of its arms -- cost
		// the finishing arm a copy of every register too, ~65 moves per slot instead of 22)
		A = B;
		ckA = ckB;
		if (busy)
			fetch(B, ckB);
		k3_store_t<WHB>(k3_tile, dst);
	}
	if (MODE == 1) {  // (one atomic per wave: one per segment -- 50 k a launch on one address -- queues up in the L2, see stat_flush)
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1)
			stat_slots += __shfl_xor(stat_slots, o, 64);
#ifndef TFREC_AMD_PROFILE_WHB  // (that build counts the WHB demodulator's cycles in this slot)
		if ((threadIdx.x & 63) == 0 && stat_slots)
			atomicAdd(&T.stats[5], stat_slots);
#endif
	}
}

// K3c: see the K3 header.  Wave per chain.  The check of segment k -- "the last run that wrote k started from the true
// state after k-1" -- only needs table entries once k-1 is known to be good, so all segments are checked at once, one
// per lane; normally every check passes and the chain's new state is the last segment's end.  From the first segment
// that fails, lane 0 walks on alone: a flat loop that per iteration either checks one segment or repairs one slot.
// (As a lane-per-chain walk the kernel was a string of ~46 dependent table reads per chain; what remains of its time is
// the longest serial repair of the batch -- a segment that did not converge behind one that did not either.)
template <bool WHB>
__device__ __forceinline__ void fix_chain(int a, int s, int n_streams, int M, const uint32_t *__restrict__ dec,
					  size_t dec_stride, const int16_t *__restrict__ fmdev, size_t fmdev_stride,
					  const ChainLaunch &L, const WinTables &T, int16_t *__restrict__ ld16,
					  int32_t *__restrict__ dev32, int lane)
{
	const int c = a * n_streams + s;
	ChainState &st = L.states[a][s];
	const int vtotal = T.vtotal[c];
	if (vtotal == 0)
		return;
	const int nseg = (vtotal + kSegSlots - 1) / kSegSlots;
	if (lane == 0)
		atomicAdd(&T.stats[0], (unsigned long long)nseg);
	const int count = T.count[c];
	const BiquadCoef cf = L.params[a].iir;
	const BiquadEnd *e1 = T.segend1 + (size_t)c * T.segcap;
	const void *in = WHB ? (const void *)(dec + (size_t)s * dec_stride) : (const void *)(fmdev + (size_t)s * fmdev_stride);
	void *out = WHB ? (void *)(dev32 + (size_t)s * T.slots * 32) : (void *)(ld16 + (size_t)(c - T.ld_c0) * T.slots * 32);
	const uint32_t prev0 = T.prevdec[s];
	const double2 *ckrow = T.ckpt + (size_t)(c - T.ck_c0) * T.slots;
	// the end state of segment kk, IF the last run that wrote it started from the true state
	auto end_if_good = [&](int kk) -> BiquadEnd {
		// (segment 0: its repair run K3b started from the carried state, which is the true one)
		const size_t sk = (size_t)c * T.segcap + kk;
		const int fx2 = T.segfix2[sk];
		const bool second = (fx2 & kSegRan) != 0;
		const int fx = second ? fx2 : T.segfix[sk];
		return (fx & kSegConverged) ? e1[kk] : (second ? T.segend3[sk] : T.segend2[sk]);
	};
	int k = nseg;  // the first segment whose last run did not start from the end of its predecessor
	for (int base = 1; base < nseg; base += 64) {
		const int kk = base + lane;
		bool bad = false;
		if (kk < nseg) {
			const size_t sk = (size_t)c * T.segcap + kk;
			const bool second = (T.segfix2[sk] & kSegRan) != 0;
			const BiquadEnd from = second ? T.segend2[sk - 1] : e1[kk - 1];
			const BiquadEnd t = end_if_good(kk - 1);
			bad = !(same_bits(t.yn, from.yn) && same_bits(t.yn1, from.yn1) && same_bits(t.dn1, from.dn1) && same_bits(t.dn2, from.dn2));
		}
		const unsigned long long any = __ballot(bad);
		if (any) {
			k = base + __builtin_ctzll(any);
			break;
		}
	}
	if (lane != 0)
		return;
	Biquad f = biquad_of(end_if_good(k - 1));  // the TRUE state after segment k - 1
	BiquadEnd prev = e1[k - 1], cur = prev;
	bool repairing = false;
	// repair state
	int j = 0, i = 0, nslots = 0, min_slots = 0, done = 0, nsamples = 0;
	SegWin cw = { 0, 0, 0, 0 }, nw = cw;
	K3Chunk<WHB> A, B;
	double2 ckA = make_double2(0, 0), ckB = ckA;
	while (true) {
		if (!repairing) {
			if (k >= nseg)
				break;
			cur = e1[k];
			const size_t sk = (size_t)c * T.segcap + k;
			const int fx2 = T.segfix2[sk];
			const bool second = (fx2 & kSegRan) != 0;  // the LAST run that wrote segment k: K3b' or K3b
			const int fx = second ? fx2 : T.segfix[sk];
			const BiquadEnd from = second ? T.segend2[sk - 1] : prev;  // the state that run started from
			if (same_bits(f.yn, from.yn) && same_bits(f.yn1, from.yn1) && same_bits(f.dn1, from.dn1) &&
			    same_bits(f.dn2, from.dn2)) {
				// it ran segment k from the true state: what is stored now is the true trajectory
				f = biquad_of((fx & kSegConverged) ? cur : (second ? T.segend3[sk] : T.segend2[sk]));
				prev = cur;
				k++;
				continue;
			}
			// it started from a wrong state: repair serially from f, at least as far as it had written
			atomicAdd(&T.stats[2], 1ull);
			const uint2 start = T.segstart[(size_t)c * T.segcap + k];
			j = (int)start.x;
			i = (int)start.y;
			const int left = vtotal - k * kSegSlots;
			nslots = left < kSegSlots ? left : kSegSlots;
			min_slots = fx & ~(kSegConverged | kSegRan);
			done = 0;
			nsamples = 0;
			cw = seg_win(T, c, j, M);
			nw = seg_win(T, c, j + 1 < count ? j + 1 : j, M);
			k3_load<WHB>(A, in, cw.og + kChunk * i, prev0);
			ckA = ckrow[cw.slot0 + i];  // (read for every slot, used at checkpoint slots only)
			repairing = true;
		}
		// one slot of the repair run (cf. seg_run<.., true>)
		const bool hop = i + 1 >= cw.nch;
		const bool more = done + 1 < nslots;
		const int og2 = hop ? nw.og : cw.og, i2 = hop ? 0 : i + 1, slot2 = (hop ? nw.slot0 : cw.slot0) + i2;
		if (more) {
			k3_load<WHB>(B, in, og2 + kChunk * i2, prev0);
			ckB = ckrow[slot2];
		}
		uint32_t ow[WHB ? 32 : 16];
		const int nv = cw.n - kChunk * i < kChunk ? cw.n - kChunk * i : kChunk;
		k3_filter<WHB>(f, cf, A, nv, ow);
		k3_store<WHB>(out, cw.slot0 + i, ow);
		nsamples += nv;
		done++;
		const bool joined = ck_slot(cw.slot0 + i) && same_bits(f.yn, ckA.x) && same_bits(f.yn1, ckA.y) && nsamples >= 2 &&
				    done >= min_slots;
		if (joined || !more) {
			if (joined)
				f = biquad_of(cur);  // joined the speculative trajectory: its end state is the true one
			prev = cur;
			k++;
			repairing = false;
			continue;
		}
		if (hop) {
			j++;
			cw = nw;
			nw = seg_win(T, c, j + 1 < count ? j + 1 : j, M);
		}
		i = i2;
		A = B;
		ckA = ckB;
	}
	st.iir = f;
}

__global__ __launch_bounds__(64) void fix_biquad_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							const int16_t *__restrict__ fmdev, size_t fmdev_stride, int n_streams,
							int n_blocks, ChainLaunch L, WinTables T, int16_t *__restrict__ ld16,
							int32_t *__restrict__ dev32, int want_kind)
{
	__builtin_amdgcn_s_setprio(3);  // see windows_kernel
	const int a = blockIdx.y;
	const int s = blockIdx.x;  // wave per chain
	const int M = n_blocks * kBlockDec;
	const int kind = L.params[a].kind;
	if (kind != want_kind)
		return;
	if (kind == 1)
		fix_chain<false>(a, s, n_streams, M, dec, dec_stride, fmdev, fmdev_stride, L, T, ld16, dev32, (int)threadIdx.x);
	else if (kind == 2)
		fix_chain<true>(a, s, n_streams, M, dec, dec_stride, fmdev, fmdev_stride, L, T, ld16, dev32, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------ slicers

struct Slicer {  // window-local demodulator state (tfa1.h:28-32, tfa2.h:35-42)
	int lbi;     // last_bit_idx, relative to block cur_block
	int cur_block;
	int mark_lvl, rssi_i;                          // tfa1 (rssi_i also tfa2)
	int bitcnt, dmin, dmax, offset, last_bit;      // tfa2
	int first_cand_g;
	int td_lo, td_hi;  // tfa2.cpp:393 "tdiff > spb / 4 && tdiff < 32 * spb" for the integer tdiff: td_lo <= tdiff <= td_hi
	// The lane-per-window loop walks a window in 32-sample chunks.  demodulator::start (decoder.cpp:118-122) rebases
	// last_bit_idx at every block start; a chunk holds at most one block start, at its sample `split` (>= 32: none): the
	// per-sample form of this bookkeeping (block of the sample, compare, rebase, index) was a fifth of a sample's instructions
	int ib;     // index (decoder.h:72 units: 2 per sample) of the chunk's first sample relative to block cur_block
	int split;  // sample of the chunk at which block cur_block + 1 begins
	int hi, lo;  // tfa2.cpp:379-381: noffset + dmax / 32, noffset + dmin / 32 -- functions of (offset, dmax, dmin), which only move
	             // while bitcnt < 10: kept instead of recomputed at every sample (a conversion to double and back, a product
	             // and two range compares per sample of a loop that runs at a lone wave's issue rate)
};
__device__ __forceinline__ void tfa2_thresholds(Slicer &f)
{
	const int noffset = d2i(0.9 * f.offset);
	f.hi = noffset + f.dmax / 32;
	f.lo = noffset + f.dmin / 32;
}

__device__ __forceinline__ void slicer_fresh(Slicer &f, int kind)
{
	f.mark_lvl = 0;
	f.rssi_i = 0;
	f.bitcnt = 0;
	f.dmin = 32767;
	f.dmax = -32767;
	f.offset = 0;
	f.last_bit = 0;
	f.first_cand_g = -1;
	f.hi = f.lo = 0;
	(void)kind;
}

// the chunk that begins at sample gf: last_bit_idx to the block of its first sample, where the next block begins in it
__device__ __forceinline__ void slicer_chunk_begin(Slicer &f, int gf)
{
	const int b = gf >> 13;
	if (b != f.cur_block) {
		f.lbi = rebase_lbi(f.lbi, f.cur_block, b);
		f.cur_block = b;
	}
	const int o = gf & (kBlockDec - 1);
	f.ib = 2 * o;
	f.split = kBlockDec - o;
}
// sample k of the chunk: its index; crossing into the next block is rare and tested for the whole wave at once
__device__ __forceinline__ int slicer_index(Slicer &f, int k)
{
	if (__builtin_expect(__ballot(k == f.split) != 0ull, 0)) {
		if (k == f.split) {
			f.lbi = rebase_lbi(f.lbi, f.cur_block, f.cur_block + 1);
			f.cur_block++;
			f.ib -= kIndexSpan;
		}
	}
	return f.ib + 2 * k;
}

// One sample of tfa1_demod::demod inside a window (tfa1.cpp:150-178); the flush at the window's last sample
// is done by the caller.  (BITPERIOD 10: ones are emitted for n = 22, 42, ... <= gap.)
__device__ __forceinline__ void tfa1_sample(Slicer &f, BitWriter &bw, int k, int I, int Q, int pI, int pQ)
{
	const int index = slicer_index(f, k);
	const int dev = fm_dev_nrzs(I, Q, pI, pQ);
	{  // (both sides evaluated, then selected: as a branch the decay cost the wave three scalar mask instructions per sample)
		const int decayed = tfa1_decay(f.mark_lvl);
		f.mark_lvl = dev > f.mark_lvl ? dev : decayed;
	}
	if (f.mark_lvl > f.rssi_i)
		f.rssi_i = f.mark_lvl;
	if (dev < (int)((uint32_t)f.mark_lvl >> 1)) {  // mark_lvl / 2 (tfa1.cpp:164): mark_lvl >= 0, it only becomes a larger dev or its own decay
		if (f.lbi) {
			const int gap = index - f.lbi;
			if (gap > 4) {
				for (int n = 22; n <= gap; n += 20)
					bw.put(1);
				bw.put(0);
			}
		}
		if (index - f.lbi > 2)
			f.lbi = index;
	}
}

// A candidate edge at sample g (tfa2.cpp:383-411: outside the dead band, bit != last_bit): glitch rule, edge timing, the
// bits it emits, last_bit_idx.  (The caller has brought last_bit_idx to g's block.)
__device__ __forceinline__ void tfa2_candidate(Slicer &f, BitWriter &bw, int g, int index, int bit, double spb, uint64_t nb_mul)
{
	if (f.first_cand_g < 0)
		f.first_cand_g = g;
	if (index > f.lbi + 8) {
		f.bitcnt++;
		const int tdiff = index - f.lbi;
		if (tdiff >= f.td_lo && tdiff <= f.td_hi) {  // tdiff > spb / 4 && tdiff < 32 * spb
			const int numbits = nb_mul ? tfa2_numbits_mul(tdiff, nb_mul) : d2i(((tdiff / 2) + (spb / 2)) / spb);
			// numbits - 1 copies of last_bit (none if numbits >= 32: tfa2.cpp:400), then the new bit: one append
			const int run = (numbits < 32 && numbits > 1) ? numbits - 1 : 0;
			bw.put_bits((f.last_bit ? (1u << run) - 1u : 0u) | ((uint32_t)bit << run), run + 1);
			f.last_bit = bit;
		}
	}
	if (index - f.lbi > 2)
		f.lbi = index;
}

// One sample of tfa2_demod::demod inside a window (tfa2.cpp:357-412), ld = (int)iir->step(fm_dev(...)).
// iq: the decimated sample itself (looked at while 4 < bitcnt < 10 only: tfa2.cpp:371-375)
__device__ __forceinline__ void tfa2_sample(Slicer &f, BitWriter &bw, int g, int k, int ld, uint32_t iq, double spb, uint64_t nb_mul)
{
	const int index = slicer_index(f, k);
	if (f.bitcnt < 10) {
		const bool up = ld > f.dmax, down = ld < f.dmin;
		if (up)
			f.dmax = (7 * f.dmax + ld) / 8;
		if (down)
			f.dmin = (7 * f.dmin + ld) / 8;
		if (up || down) {  // offset and the thresholds are functions of (dmax, dmin): tfa2.cpp:369, 379-381
			f.offset = (f.dmax + f.dmin) / 2;
			tfa2_thresholds(f);
		}
		if (f.bitcnt > 4) {  // wrapping int32 arithmetic as in the reference binary (tfa2.cpp:373)
			const int I = (int)(int16_t)(iq & 0xffff), Q = (int)iq >> 16;
			const uint32_t t = (uint32_t)f.rssi_i + (uint32_t)(I * I) + (uint32_t)(Q * Q);
			f.rssi_i = (int)((uint32_t)f.rssi_i + (uint32_t)((int)t / 100));
		}
	}
	const int hi = f.hi, lo = f.lo;
	const int bit = ld > hi ? 1 : 0;
	if ((ld > hi || ld < lo) && bit != f.last_bit)
		tfa2_candidate(f, bw, g, index, bit, spb, nb_mul);
}

// Run one window [g0, last] of a TFA_1 (KIND 0) or TFA_2-family (KIND 1) slicer.  `f` carries the state in and
// out; bits go to bw.  Returns with f.cur_block = block of `last`.
// plain-value register blocks for prefetching (arrays behind references end up in scratch)
struct Slot8 {
	uint4 q0, q1, q2, q3, q4, q5, q6, q7;
};
struct Slot4 {
	uint4 q0, q1, q2, q3;
};

// Run one window [g0, last] of a TFA_1 (KIND 0) or TFA_2-family (KIND 1) slicer.  Each 32-sample chunk is moved
// from registers to the lane's LDS column, the next chunk's loads are issued, then the chunk is walked from
// LDS by a rolled loop (small code, HBM latency overlapped with the state machine).
// head_chunks > 0 (TFA_2 family, long windows): stop after the chunk in which bitcnt reached 10 (the thresholds
// are frozen from there on), at the latest after head_chunks chunks; the wave-cooperative slicer takes over.
// Returns the first chunk NOT done (nch: all).
template <int KIND>
__device__ __forceinline__ int run_window(Slicer &f, BitWriter &bw, int g0, int last, bool closed,
					  const uint32_t *__restrict__ drow, const uint32_t *__restrict__ ldslots, int prevI,
					  int prevQ, double spb, uint64_t nb_mul, uint4 *__restrict__ my_lds, int head_chunks)
{
	const int n = last - g0 + 1;
	const int nch = (n + kChunk - 1) >> 5;
	if (KIND == 0) {
		auto load = [&](int i) -> Slot8 {
			const u32x4_a4 *p = reinterpret_cast<const u32x4_a4 *>(drow + g0 + kChunk * i);
			Slot8 r;
			u32x4_a4 v;
			v = p[0]; r.q0 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[1]; r.q1 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[2]; r.q2 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[3]; r.q3 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[4]; r.q4 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[5]; r.q5 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[6]; r.q6 = make_uint4(v.x, v.y, v.z, v.w);
			v = p[7]; r.q7 = make_uint4(v.x, v.y, v.z, v.w);
			return r;
		};
		int pI = prevI, pQ = prevQ;
		if (g0 > 0) {
			const uint32_t pw = drow[g0 - 1];
			pI = (int)(int16_t)(pw & 0xffff);
			pQ = (int)pw >> 16;
		}
		Slot8 cur = load(0);
		for (int i = 0; i < nch; i++) {
			my_lds[0 * 64] = cur.q0; my_lds[1 * 64] = cur.q1; my_lds[2 * 64] = cur.q2; my_lds[3 * 64] = cur.q3;
			my_lds[4 * 64] = cur.q4; my_lds[5 * 64] = cur.q5; my_lds[6 * 64] = cur.q6; my_lds[7 * 64] = cur.q7;
			const Slot8 nxt = load(i + 1 < nch ? i + 1 : i);
			const int nv = n - kChunk * i < kChunk ? n - kChunk * i : kChunk;
			slicer_chunk_begin(f, g0 + kChunk * i);
			uint4 vn = my_lds[0];
#pragma unroll 1
			for (int q = 0; 4 * q < nv; q++) {
				const uint4 v = vn;
				vn = my_lds[((q + 1) & 7) * 64];
				const uint32_t vw[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int t = 0; t < 4; t++) {
					const int I = (int)(int16_t)(vw[t] & 0xffff), Q = (int)vw[t] >> 16;
					if (4 * q + t < nv)
						tfa1_sample(f, bw, 4 * q + t, I, Q, pI, pQ);
					pI = I;
					pQ = Q;
				}
			}
			bw.chunk_end();
			cur = nxt;
		}
	} else {
		// ld = biquad output, window-relative slots of 32 samples (K3)
		auto load = [&](int i) -> Slot4 {
			const uint4 *p = reinterpret_cast<const uint4 *>(ldslots + (size_t)i * 16);
			Slot4 r;
			r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2]; r.q3 = p[3];
			return r;
		};
		Slot4 cur = load(0);
		for (int i = 0; i < nch; i++) {
			my_lds[0 * 64] = cur.q0; my_lds[1 * 64] = cur.q1; my_lds[2 * 64] = cur.q2; my_lds[3 * 64] = cur.q3;
			const Slot4 nxt = load(i + 1 < nch ? i + 1 : i);
			const int nv = n - kChunk * i < kChunk ? n - kChunk * i : kChunk;
			slicer_chunk_begin(f, g0 + kChunk * i);
			uint4 vn = my_lds[0];
#pragma unroll 1
			for (int q = 0; 8 * q < nv; q++) {
				const uint4 v = vn;
				vn = my_lds[((q + 1) & 3) * 64];
				const uint32_t vw[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int t = 0; t < 8; t++) {
					if (8 * q + t < nv) {
						const int ld = (int)(int16_t)((vw[t >> 1] >> (16 * (t & 1))) & 0xffff);
						// sample 8 q + t of the chunk: group 2 q + (t >> 2) of four, component t & 3
						// (the sample itself is looked at while 4 < bitcnt < 10 only: tfa2.cpp:371-375.  Staging the chunk's 32 samples in
						// LDS instead of this load-and-wait made the slicers 20 % faster and the batch 3 % slower: profiles/NOTES.md round 3)
						const uint32_t iq = (f.bitcnt > 4 && f.bitcnt < 10) ? drow[g0 + kChunk * i + 8 * q + t] : 0u;
						tfa2_sample(f, bw, g0 + kChunk * i + 8 * q + t, 8 * q + t, ld, iq, spb, nb_mul);
					}
				}
			}
			bw.chunk_end();
			cur = nxt;
			if (head_chunks > 0 && (f.bitcnt >= 10 || i + 1 >= head_chunks) && i + 1 < nch)
				return i + 1;
		}
	}
	const int bl = last >> 13;
	if (bl != f.cur_block) {
		f.lbi = rebase_lbi(f.lbi, f.cur_block, bl);
		f.cur_block = bl;
	}
	if (closed && KIND == 1)  // tfa2.cpp:430-431: trailing bits before the flush
		for (int q = 0; q < 16; q++)
			bw.put(f.last_bit);
	return nch;
}

template <int KIND>
__device__ __forceinline__ void window_task(int c, int j, int n_streams, int M, const uint32_t *__restrict__ dec,
					    size_t dec_stride, const int16_t *__restrict__ ld16, const ChainLaunch &L,
					    const WinTables &T, bool exact_lbi, int lbi_in_override, uint4 *__restrict__ my_lds,
					    int head_chunks)
{
	const int a = c / n_streams, s = c - a * n_streams;
	const ChainParams &p = L.params[a];
	const ChainState &st = L.states[a][s];
	const int og = T.open[(size_t)c * T.cap + j];
	const int close = T.close[(size_t)c * T.cap + j];
	const bool closed = close < M;
	const int last = closed ? close : M - 1;
	const bool cont = (j == 0) && T.cont[c];
	Slicer f;
	slicer_fresh(f, KIND);
	f.cur_block = og >> 13;
	if (cont) {  // resume the window the previous submit left open
		f.mark_lvl = st.mark_lvl;
		f.rssi_i = st.rssi_i;
		f.bitcnt = st.bitcnt;
		f.dmin = st.dmin;
		f.dmax = st.dmax;
		f.offset = st.offset;
		f.last_bit = st.last_bit;
		f.lbi = rebase_lbi(st.last_bit_idx, -1, f.cur_block);
	} else if (KIND == 0) {
		f.lbi = 0;  // tfa1.cpp:183
	} else if (exact_lbi) {
		f.lbi = lbi_in_override;
	} else if (j == 0) {
		f.lbi = rebase_lbi(st.last_bit_idx, -1, f.cur_block);  // known exactly: carried state
	} else {
		f.lbi = kSpecLbi;  // speculation, validated by commit_kernel
	}
	if (KIND == 1) {
		tfa2_thresholds(f);
		f.td_lo = (int)floor(p.spb / 4) + 1;
		f.td_hi = (int)ceil(32 * p.spb) - 1;
	}
	BitWriter bw{ T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j, 0u, 0, 0u, -1 };
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const uint32_t *ldslots = (KIND == 1) ? reinterpret_cast<const uint32_t *>(ld16 + (size_t)(c - T.ld_c0) * T.slots * 32) +
							(size_t)win_slot0(og, j) * 16
					      : nullptr;
	const int resume = run_window<KIND>(f, bw, og, last, closed, drow, ldslots, st.prev_i, st.prev_q, p.spb, p.nb_mul, my_lds,
					    head_chunks);
	bw.finish();
	WinResult &r = T.result[(size_t)c * T.cap + j];
	r.resume = resume < ((last - og + 1 + kChunk - 1) >> 5) ? resume : -1;
	r.nbits = bw.n;
	r.closed = closed ? 1 : 0;
	r.rssi_i = f.rssi_i;
	r.offset = f.offset;
	r.lbi_out = f.lbi;
	r.first_cand_g = f.first_cand_g;
	r.bitcnt = f.bitcnt;
	r.dmin = f.dmin;
	r.dmax = f.dmax;
	r.last_bit = f.last_bit;
	r.mark_lvl = f.mark_lvl;
}

// ------------------------------------------------------------------------------------------------ K4
// Lane per window.  blockIdx.y = protocol kind (0 TFA_1, 1 TFA_2 family), so a wave runs one slicer type.  Short
// windows are sliced completely.  Of the long TFA_2-family windows only the head, where the thresholds still
// adapt sample by sample (tfa2.cpp:363 "bitcnt < 10"; cheap per window when 64 windows share a wave, expensive
// for a whole wave) -- the rest, and the long TFA_1 windows, belong to coop_slicer_kernel.
__global__ __launch_bounds__(64) TFREC_LAT_VGPR_ATTR void slicer_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						    const int16_t *__restrict__ ld16, int n_streams, int n_blocks, ChainLaunch L,
						    WinTables T, int lanes, int head_chunks, int kind, int qsel)
{
	// qsel: 0 = long windows (heads), then short ones; 1 = only the long windows' heads; 2 = only the short windows
	// the lanes' 32-sample chunk, a column each: 8 KB for TFA_1 (32 dwords per lane), 4 KB for the TFA_2 family (32 int16).
	// Dynamic, so that the TFA_2-family launch holds half: these waves live for milliseconds, six of them per CU, and the
	// front end beside them needs 16.6 KB per workgroup of what the CU's 160 KB have left (profiles/NOTES.md round 3)
	extern __shared__ uint4 slot_lds[];
	latency_prio();
#ifdef TFREC_AMD_SLICER_CLAIM  // (sensitivity experiment: -DTFREC_AMD_SLICER_CLAIM='"v175"' makes the kernel hold that many registers)
	asm volatile("" ::: TFREC_AMD_SLICER_CLAIM);
#endif
	uint4 *my_lds = slot_lds + threadIdx.x;
	if ((int)threadIdx.x >= lanes)
		return;
	const int M = n_blocks * kBlockDec;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	for (int q = 2 * kind + (kind == 0 ? 1 : 0); q < 2 * kind + 2; q++) {
		if ((qsel == 1 && (q & 1)) || (qsel == 2 && !(q & 1)))
			continue;
		const uint32_t count = T.queue[q].count;
		const int head = (q & 1) == 0 ? head_chunks : 0;
		while (true) {
			const uint32_t idx = atomicAdd(&T.queue[q].head, 1u);
			if (idx >= count)
				break;
			const uint2 it = T.items[(size_t)q * total + idx];
			const int c = (int)it.x, j = (int)it.y;
			if (kind == 0)
				window_task<0>(c, j, n_streams, M, dec, dec_stride, ld16, L, T, false, 0, my_lds, 0);
			else
				window_task<1>(c, j, n_streams, M, dec, dec_stride, ld16, L, T, false, 0, my_lds, head);
		}
	}
}

// ------------------------------------------------------------------------------------------------ K4a' TFA_1 marks
// The TFA_1 peak detector mark_lvl = dev > mark_lvl ? dev : (int)(mark_lvl * 0.95) (tfa1.cpp:157-160) is a serial
// recurrence, but a forgetful one: at every sample with dev > mark_lvl the state becomes dev whatever it was.
// Lane per PIECE of 1024 samples of a long window: the lane starts 256 samples early from mark_lvl = 0 (the
// window's first piece from the true initial value), and stores for its piece the bits "dev < mark_lvl / 2"
// (tfa1.cpp:164), the maximum (rssi) and the value before / after the piece.  coop_slicer_kernel checks
// start == the true value bit for bit when it reaches the piece, and otherwise recomputes the piece itself:
// exactness does not rest on the warm-up, only speed does.  16 lane-instructions per sample for 64 pieces at once
// instead of 7 wave-instructions per sample.
__global__ __launch_bounds__(256) TFREC_LAT_VGPR_ATTR void mark_kernel(const uint32_t *__restrict__ dec, size_t dec_stride, int n_streams,
						  int n_blocks, ChainLaunch L, WinTables T)
{
	latency_prio();
	const int M = n_blocks * kBlockDec;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	const uint32_t count = T.queue[7].count;
	// (four independent waves per workgroup, one on each SIMD of a CU: see whb_verify_kernel)
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
	for (uint32_t idx = tid; idx < count; idx += nthreads) {
		const uint2 it = T.items[(size_t)7 * total + idx];
		const int c = (int)it.x, j = (int)(it.y & 0x1ffffu), pc = (int)(it.y >> 17);
		const int a = c / n_streams, s = c - a * n_streams;
		const ChainState &st = L.states[a][s];
		const int og = T.open[(size_t)c * T.cap + j];
		const int close = T.close[(size_t)c * T.cap + j];
		const int n = (close < M ? close : M - 1) - og + 1;
		const uint32_t *drow = dec + (size_t)s * dec_stride;
		const uint32_t prev0 = ((uint32_t)st.prev_i & 0xffffu) | ((uint32_t)st.prev_q << 16);
		const int slot0 = win_slot0(og, j);
		const int i0 = pc * kMarkSlots;                                          // first slot of the piece
		const int nch = (n + 31) >> 5;
		const int i1 = nch < i0 + kMarkSlots ? nch : i0 + kMarkSlots;
		const int iw = pc == 0 ? 0 : i0 - kMarkWarmSlots;                      // warm-up start (pc >= 1: i0 >= 32)
		int mark = (pc == 0 && j == 0 && T.cont[c]) ? st.mark_lvl : 0;
		int start = mark, mx = 0;
		K3Chunk<true> A, B;
		k3_load<true>(A, drow, og + 32 * iw, prev0);
		for (int i = iw; i < i1; i++) {
			if (i + 1 < i1)
				k3_load<true>(B, drow, og + 32 * (i + 1), prev0);
			if (i == i0)
				start = mark;
			const int nv = n - 32 * i < 32 ? n - 32 * i : 32;
			int pI = (int)(int16_t)(A.prevw & 0xffff), pQ = (int)A.prevw >> 16;
			uint32_t bits = 0;
			if (__ballot(nv < 32) == 0ull) {
				// A whole chunk in every lane (all but a window's last): no per-sample guard, the decay computed beside the
				// compare instead of under a mask, mark_lvl / 2 as a shift (mark_lvl >= 0: it starts at 0 and only ever becomes
				// a larger dev or its own decay), the bits shifted in by an add-with-carry, and the maximum taken over dev:
				// max_k mark_k = max(mark_0, max_{k >= 1} dev_k) -- no mark exceeds that, and the largest dev either becomes
				// the mark or meets one that is no smaller.  13.5 vector instructions per sample instead of 19 + 5 scalar.
				uint32_t rev = 0;
				int dmax = -0x7fffffff;
#pragma unroll
				for (int k = 0; k < 32; k++) {
					const int I = (int)(int16_t)(A.w[k] & 0xffff), Q = (int)A.w[k] >> 16;
					const int dev = fm_dev_nrzs(I, Q, pI, pQ);
					const int decayed = tfa1_decay(mark);
					mark = dev > mark ? dev : decayed;
					const int half = (int)((uint32_t)mark >> 1);
					asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rev) : "v"(dev), "v"(half) : "vcc");
					if (k == 0)
						mx = mark > mx ? mark : mx;
					else
						dmax = dev > dmax ? dev : dmax;
					pI = I;
					pQ = Q;
				}
				mx = dmax > mx ? dmax : mx;
				bits = __builtin_bitreverse32(rev);
			} else {
#pragma unroll
				for (int k = 0; k < 32; k++) {
					const int I = (int)(int16_t)(A.w[k] & 0xffff), Q = (int)A.w[k] >> 16;
					if (k < nv) {
						const int dev = fm_dev_nrzs(I, Q, pI, pQ);
						mark = dev > mark ? dev : tfa1_decay(mark);
						mx = mark > mx ? mark : mx;
						bits |= (uint32_t)(dev < mark / 2) << k;
					}
					pI = I;
					pQ = Q;
				}
			}
			if (i >= i0)
				T.cand[(size_t)s * T.slots + slot0 + i] = bits;
			else
				mx = 0;  // the warm-up does not count
			A = B;
		}
		MarkPiece mp;
		mp.start = start;
		mp.end = mark;
		mp.max = mx;
		mp.pad_ = 0;
		T.mark[(size_t)s * T.slots + slot0 + i0] = mp;
	}
}

// ------------------------------------------------------------------------------------------------ K4b
// Wave-cooperative slicers for LONG windows (kLongWindow): one wave per window.
// A lane-per-window slicer needs ~100 instructions per sample on a serial path; a 40 000-sample burst then
// takes milliseconds whatever the GPU's width.  Two forms of the same rules:
//   * the scalar walks (rounds 2-4): lane n owns sample n of a 64-sample step, the per-sample work is done by 64 lanes at
//     once and only the sparse part stays serial (wave-uniform) --
//       TFA_2 family (tfa2.cpp:357-412, after the thresholds froze): the candidate edges are two ballots
//           (ld > hi, ld < lo); the walk visits only the candidates of the polarity that can flip last_bit.
//       TFA_1 (tfa1.cpp:150-178): the peak detector mark_lvl = dev > mark_lvl ? dev : (int)(mark_lvl*0.95) is a
//           64-step uniform recurrence (6 instructions per sample); "dev < mark_lvl/2" is a ballot, and the walk
//           handles each RUN of consecutive candidates in O(1): only the first sample of a run can emit bits (later
//           gaps are <= 4), the others move last_bit_idx forward by 4 every second sample.
//     Bits are appended by a wave-uniform writer (lane 0 stores).  51-55 scalar instructions per edge / run: 0.51 G of the
//     benchmark batch's 1.14 G scalar instructions;
//   * a STEP PER LANE (round 5; coop_tfa1 / coop_tfa2's group_vec): 64 steps per pass,
//     every lane walks the candidates of its own step with the same formulas in absolute index units, the lanes' bits are
//     joined by coop_join_bits.  The scalar walks are what a group falls back to (0.7 % / 2.6 % of the groups).
struct CoopBits {
	uint32_t *base;
	unsigned long long acc;
	int nacc;  // valid bits in acc
	int n;     // bits written so far, including acc
	__device__ __forceinline__ void init(uint32_t *b, int nbits)
	{
		base = b;
		n = nbits;
		nacc = nbits & 31;
		acc = nacc ? (unsigned long long)(b[nbits >> 5] & ((1u << nacc) - 1u)) : 0ull;
	}
	__device__ __forceinline__ void put_run(int bit, int cnt)
	{
		while (cnt > 0) {
			const int take = cnt < 32 ? cnt : 32;
			if (bit)
				acc |= ((1ull << take) - 1ull) << nacc;
			nacc += take;
			n += take;
			cnt -= take;
			if (nacc >= 32) {
				if (threadIdx.x == 0)
					base[(n - nacc) >> 5] = (uint32_t)acc;
				acc >>= 32;
				nacc -= 32;
			}
		}
	}
	// cnt in [1, 32] bits at once, bit k of v = the k-th of them
	__device__ __forceinline__ void put_bits(uint32_t v, int cnt)
	{
		acc |= (unsigned long long)v << nacc;  // nacc < 32 here
		nacc += cnt;
		n += cnt;
		if (nacc >= 32) {
			if (threadIdx.x == 0)
				base[(n - nacc) >> 5] = (uint32_t)acc;
			acc >>= 32;
			nacc -= 32;
		}
	}
	__device__ __forceinline__ void finish()
	{
		if (nacc && threadIdx.x == 0)
			base[(n - nacc) >> 5] = (uint32_t)acc;
	}
};

// Join the bits the 64 lanes of a wave produced (lane l: `cnt` <= 64 bits in `acc`, LSB first; lane order = bit order) and append
// them to the wave-uniform writer: prefix sum of the counts, an LDS image of the output words from the writer's pending word
// on (three ORs per lane), whole words stored by all lanes, the rest becomes the writer's pending word.  One-wave workgroups.
constexpr int kCoopStageWords = 136;  // 31 carried bits + 64 lanes * 64 bits, + the reach of a lane's three ORs
__device__ __forceinline__ void coop_join_bits(CoopBits &bw, uint32_t *__restrict__ stage, unsigned long long acc, int cnt)
{
	const int lane = threadIdx.x;
	int incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(incl, o, 64);
		incl += lane >= o ? t : 0;
	}
	const int total = __builtin_amdgcn_readlane(incl, 63);
	if (total == 0)
		return;
	const int nacc = bw.nacc;
	for (int i = lane; i < kCoopStageWords; i += 64)
		stage[i] = (i == 0) ? (uint32_t)bw.acc : 0u;
	__syncthreads();
	if (cnt > 0) {
		const int pos = nacc + incl - cnt;
		const int sh = pos & 31, w0 = pos >> 5;
		const unsigned long long lo = acc << sh;
		const uint32_t hi = sh ? (uint32_t)(acc >> (64 - sh)) : 0u;
		if ((uint32_t)lo)
			atomicOr(&stage[w0], (uint32_t)lo);
		if ((uint32_t)(lo >> 32))
			atomicOr(&stage[w0 + 1], (uint32_t)(lo >> 32));
		if (hi)
			atomicOr(&stage[w0 + 2], hi);
	}
	__syncthreads();
	const int nw = (nacc + total) >> 5;  // completed words
	uint32_t *out = bw.base + ((bw.n - nacc) >> 5);
	for (int i = lane; i < nw; i += 64)
		out[i] = stage[i];
	const uint32_t pend = stage[nw];
	__syncthreads();
	bw.acc = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
	bw.nacc = (nacc + total) & 31;
	bw.n += total;
}

__device__ __forceinline__ void coop_tfa2(int c, int j, int n_streams, int M, const uint32_t *__restrict__ dec,
					  size_t dec_stride, const int16_t *__restrict__ ld16, const ChainLaunch &L,
					  const WinTables &T, uint32_t *__restrict__ stage, GroupStats &gs, bool fresh = false, int fresh_lbi = 0)
{
	const int lane = threadIdx.x;
	const int a = c / n_streams, s = c - a * n_streams;
	const double spb = L.params[a].spb;
	const uint64_t nb_mul = L.params[a].nb_mul;
	const int og = T.open[(size_t)c * T.cap + j];
	const int close = T.close[(size_t)c * T.cap + j];
	const bool closed = close < M;
	const int last = closed ? close : M - 1;
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const int16_t *ldrow = ld16 + (size_t)(c - T.ld_c0) * T.slots * 32 + (size_t)win_slot0(og, j) * 32;  // window-relative
	// ---- wave-uniform slicer state (tfa2.h:35-42): where the lane-per-window head (slicer_kernel) stopped
	// (fresh: the whole window from its first sample, with the given last_bit_idx -- commit's exact re-slice)
	WinResult &rr = T.result[(size_t)c * T.cap + j];
	WinResult r0 = rr;
	if (fresh) {  // tfa2.cpp:436-441 as the previous window's timeout left the demodulator
		r0.resume = 0;
		r0.nbits = 0;
		r0.rssi_i = 0;
		r0.bitcnt = 0;
		r0.dmin = 32767;
		r0.dmax = -32767;
		r0.offset = 0;
		r0.last_bit = 0;
		r0.first_cand_g = -1;
		r0.lbi_out = fresh_lbi;
	}
	if (r0.resume < 0)
		return;  // the head finished the window
	const int g1 = og + kChunk * r0.resume;  // first sample still to do
	int rssi_i = r0.rssi_i, bitcnt = r0.bitcnt, dmin = r0.dmin, dmax = r0.dmax, offset = r0.offset;
	int last_bit = r0.last_bit, first_cand_g = r0.first_cand_g;
	int cur_block = fresh ? og >> 13 : (g1 - 1) >> 13;
	int lbi = r0.lbi_out;  // relative to cur_block (run_window leaves it relative to the block of its last sample)
	// integer form of "tdiff > spb / 4 && tdiff < 32 * spb" (tdiff is an integer)
	const int td_lo = L.params[a].td_lo, td_hi = L.params[a].td_hi;  // (from the kernel arguments: scalars, like the walk that uses them)
	int hi = 0, lo = 0;
	auto thresholds = [&]() {  // tfa2.cpp:379-381
		const int noffset = d2i(0.9 * offset);
		hi = noffset + dmax / 32;
		lo = noffset + dmin / 32;
	};
	thresholds();
	CoopBits bw;
	bw.init(T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j, r0.nbits);
	// one candidate edge (tfa2.cpp:383-411)
	auto candidate = [&](int g, int bit) {
		const int b = g >> 13;
		if (b != cur_block) {
			lbi = rebase_lbi(lbi, cur_block, b);
			cur_block = b;
		}
		const int index = 2 * (g & (kBlockDec - 1));
		if (first_cand_g < 0)
			first_cand_g = g;
		if (index > lbi + 8) {  // tfa2.cpp:391-406
			bitcnt++;
			const int tdiff = index - lbi;
			if (tdiff >= td_lo && tdiff <= td_hi) {  // tdiff > spb / 4 && tdiff < 32 * spb
				const int numbits = nb_mul ? tfa2_numbits_mul(tdiff, nb_mul) : d2i(((tdiff / 2) + (spb / 2)) / spb);
				if (numbits < 32)
					bw.put_run(last_bit, numbits - 1);
				bw.put_run(bit, 1);
				last_bit = bit;
			}
		}
		if (index - lbi > 2)
			lbi = index;
	};
	// FOUR steps' samples per load: lane l fetches samples l, 64 + l, 128 + l, 192 + l of a 256-sample stretch, the next
	// stretch's loads are issued before this one is walked.  (One step per load, its value converted where it was loaded,
	// made the wave wait out the load's full latency in EVERY step: 3500 cycles per 64-sample step for ~110 instructions
	// of work, and the longest window's 1024 steps set the kernel's time.)  The power (tfa2.cpp:371-375) is only looked at
	// while the thresholds adapt -- a head that gave up, or commit's exact re-slice: loaded where it is used.
	struct In4 {
		uint32_t l0, l1, l2, l3;  // the int16 values as loaded, zero-extended: converting (or packing) them here would be the loads' first use
	};
	auto load4 = [&](int gb4) -> In4 {
		In4 v;
		const int g0_ = gb4 + lane, g1_ = g0_ + 64, g2_ = g0_ + 128, g3_ = g0_ + 192;
		const uint16_t *lu = reinterpret_cast<const uint16_t *>(ldrow);
		v.l0 = lu[(g0_ <= last ? g0_ : last) - og];
		v.l1 = lu[(g1_ <= last ? g1_ : last) - og];
		v.l2 = lu[(g2_ <= last ? g2_ : last) - og];
		v.l3 = lu[(g3_ <= last ? g3_ : last) - og];
		return v;
	};
#ifdef TFREC_AMD_COOPSTAT
	unsigned long long cs_steps = 0, cs_acc = 0, cs_rej = 0, cs_slow = 0, cs_full = 0, cs_pop = 0, cs_cont = 0;
#endif
	// The walk over one step's candidates when the step lies in ONE block (all but one in 128): last_bit_idx is brought to
	// that block, and the rest is plain scalar arithmetic on indices relative to the step -- an accepted edge appends its
	// numbits - 1 copies of last_bit and the new bit in one go (tfa2.cpp:399-404).  Same rules as the general walk further
	// down, which keeps the steps that straddle a block boundary (and contexts without the numbits multiplier).
	auto walk_one_block = [&](const int gb, const unsigned long long m1, const unsigned long long m0) {
		const int o = gb & (kBlockDec - 1);
		const int b = gb >> 13;
		if (b != cur_block) {
			lbi = rebase_lbi(lbi, cur_block, b);
			cur_block = b;
		}
		const int ibase = 2 * o;
		unsigned long long todo = ~0ull;  // positions not yet visited
#ifdef TFREC_AMD_COOPSTAT
		cs_steps++;
		{
			const unsigned long long mm = last_bit ? m0 : m1;
			cs_full += mm == ~0ull;
			cs_pop += (unsigned long long)__builtin_popcountll(mm);
			cs_cont += (mm & 1ull) && (ibase - lbi <= 4);  // the step begins inside a run that began before it
		}
#endif
		// (the callers come here only with a candidate of the polarity that can flip last_bit in the step: the first one
		// visited is the step's -- and, once per window, the window's -- first candidate edge)
		first_cand_g = first_cand_g < 0 ? gb + __builtin_ctzll(last_bit ? m0 : m1) : first_cand_g;
		// an edge is accepted iff index - lbi > 8 (tfa2.cpp:391) and td_lo <= index - lbi <= td_hi (:393): ONE unsigned compare
		const int acc_lo = td_lo > 9 ? td_lo : 9;
		const uint32_t acc_span = (uint32_t)(td_hi - acc_lo);  // (td_hi >= 32 * 22 - 1: never below acc_lo)
		while (true) {
			const unsigned long long m = (last_bit ? m0 : m1) & todo;
			if (!m)
				break;
#ifdef TFREC_AMD_COOPSTAT
			cs_rej++;
#endif
			const int k = __builtin_ctzll(m);
			todo = ~1ull << k;
			const int index = ibase + 2 * k, d = index - lbi;
			lbi = d > 2 ? index : lbi;  // tfa2.cpp:410-411 (d was taken first: the edge's timing uses the old value)
			bitcnt += d > 8 ? 1 : 0;    // tfa2.cpp:391-392
			if ((uint32_t)(d - acc_lo) <= acc_span) {
				const int numbits = tfa2_numbits_mul(d, nb_mul);
				const int run = (numbits < 32 && numbits > 1) ? numbits - 1 : 0;
				// `run` copies of last_bit, then its complement: ones below bit `run` and a zero there, or zeros and a one
				bw.put_bits((1u << run) - (uint32_t)last_bit, run + 1);
				last_bit ^= 1;
#ifdef TFREC_AMD_COOPSTAT
				cs_acc++;
				cs_rej--;
#endif
				continue;
			}
			// not accepted: the run of candidates of the same polarity right behind it cannot be either (see below); it
			// only moves last_bit_idx, to the last sample at which "index - lbi > 2" fired
			const unsigned long long rest = m >> 1 >> k;
			const int R = __builtin_ctzll(~rest);  // candidates at k + 1 .. k + R (rest has zeros at its top)
			if (R > 0) {
				const int e = index + 2 - lbi;  // index - lbi at sample k + 1 (<= 4)
				const int t_set = e > 2 ? 1 : ((2 - e) >> 1) + 2;
				if (t_set <= R)
					lbi = index + 2 * (t_set + 2 * ((R - t_set) >> 1));
				todo = ~1ull << (k + R);
			}
		}
	};
	auto old_range = [&](const int ga, const int gz) {  // the stretches of 256 samples from ga on, below gz
	In4 nxt4 = load4(ga);
	for (int gb4 = ga; gb4 < gz && gb4 <= last; gb4 += 256) {
	const In4 cur4 = nxt4;
	if (gb4 + 256 < gz && gb4 + 256 <= last)
		nxt4 = load4(gb4 + 256);
	// A whole stretch of 256 samples with frozen thresholds inside the window and inside one block: the eight ballots first,
	// then step by step -- a step without a sample that could flip last_bit (one in two) costs a scalar select and a
	// compare; last_bit may have flipped in the step before, so the test is made in order.
	if (bitcnt >= 10 && nb_mul && gb4 + 255 <= last && (gb4 & (kBlockDec - 1)) + 256 <= kBlockDec) {
		const int l0 = (int)(int16_t)cur4.l0, l1 = (int)(int16_t)cur4.l1, l2 = (int)(int16_t)cur4.l2, l3 = (int)(int16_t)cur4.l3;
		const unsigned long long h0 = __ballot(l0 > hi), h1 = __ballot(l1 > hi), h2 = __ballot(l2 > hi), h3 = __ballot(l3 > hi);
		const unsigned long long w0 = __ballot(l0 < lo) & ~h0, w1 = __ballot(l1 < lo) & ~h1, w2 = __ballot(l2 < lo) & ~h2,
					 w3 = __ballot(l3 < lo) & ~h3;
		if ((last_bit ? w0 : h0) != 0ull)
			walk_one_block(gb4, h0, w0);
		if ((last_bit ? w1 : h1) != 0ull)
			walk_one_block(gb4 + 64, h1, w1);
		if ((last_bit ? w2 : h2) != 0ull)
			walk_one_block(gb4 + 128, h2, w2);
		if ((last_bit ? w3 : h3) != 0ull)
			walk_one_block(gb4 + 192, h3, w3);
		continue;
	}
#pragma unroll 1
	for (int q4 = 0; q4 < 4; q4++) {
		const int gb = gb4 + 64 * q4;
		if (gb > last)
			break;
		const int ld = (int)(int16_t)(q4 == 0 ? cur4.l0 : (q4 == 1 ? cur4.l1 : (q4 == 2 ? cur4.l2 : cur4.l3)));
		const int nv = last - gb + 1 < 64 ? last - gb + 1 : 64;
		if (bitcnt >= 10) {  // thresholds frozen: two ballots, then only the edges of the polarity that can flip last_bit
			unsigned long long m1 = __ballot(ld > hi), m0 = __ballot(ld < lo);
			if (nv < 64) {  // the window's last step (the lanes behind its end hold the last sample again)
				const unsigned long long vm = (1ull << nv) - 1ull;
				m1 &= vm;
				m0 &= vm;
			}
			m0 &= ~m1;
			// Two steps in three hold no sample that could flip last_bit (0.66 candidates per step on the benchmark's windows):
			// nothing of the state moves then -- last_bit_idx is brought to a block where a candidate looks at it.
			if ((last_bit ? m0 : m1) == 0ull)
				continue;
			const int o = gb & (kBlockDec - 1);
			if (nb_mul && o + nv <= kBlockDec) {
				walk_one_block(gb, m1, m0);
				continue;
			}
			unsigned long long todo = ~0ull;  // positions not yet visited
#ifdef TFREC_AMD_COOPSTAT
			cs_slow++;
#endif
			while (true) {
				const unsigned long long m = (last_bit ? m0 : m1) & todo;
				if (!m)
					break;
				const int k = __builtin_ctzll(m);
				todo = k >= 63 ? 0ull : (~0ull << (k + 1));
				const int lb0 = last_bit;
				candidate(gb + k, last_bit ^ 1);
				// A RUN of candidates of the same polarity right behind a candidate that did not flip last_bit (a glitch, or
				// an edge out of the timing window; the other protocols' bursts and noise produce them every few samples):
				// none of them can be accepted.  After sample k, index - lbi is at most 4 at the next sample, grows by 2 per
				// sample and falls back to 0 whenever it exceeds 2 ("if (index - lbi > 2) lbi = index", tfa2.cpp:410-411): it
				// never exceeds 8 (:391), so the run only moves last_bit_idx -- to the last sample at which that rule fired.
				// O(1) instead of a walk over every sample of the run (within one block: the indices restart at a block's start).
				if (last_bit == lb0 && k < 63) {
					const unsigned long long rest = m >> (k + 1);
					int R = rest == ~0ull ? 63 - k : __builtin_ctzll(~rest);  // candidates at k+1 .. k+R
					const int room = (kBlockDec - 1) - ((gb + k) & (kBlockDec - 1));  // samples left in this block
					R = R < room ? R : room;
					if (R > 0) {
						const int index_k = 2 * ((gb + k) & (kBlockDec - 1));
						const int e = index_k + 2 - lbi;  // index - lbi at sample k + 1 (<= 4)
						const int t_set = e > 2 ? 1 : ((2 - e) >> 1) + 2;  // first sample of the run at which the rule fires
						if (t_set <= R)
							lbi = index_k + 2 * (t_set + 2 * ((R - t_set) >> 1));
						todo = k + R >= 63 ? 0ull : (~0ull << (k + R + 1));
					}
				}
			}
			continue;
		}
		// I*I + Q*Q in the wrapping arithmetic of the reference binary (tfa2.cpp:373; only this, the adaptive phase, looks at it)
		const bool valid = lane < nv;
		const uint32_t iq_ = drow[gb + lane <= last ? gb + lane : last];
		const int I = (int)(int16_t)(iq_ & 0xffff), Q = (int)iq_ >> 16;
		const uint32_t pw = (uint32_t)(I * I) + (uint32_t)(Q * Q);
		int pos = 0;
		while (pos < nv) {
			const unsigned long long rest = ~0ull << pos;
			// next candidate edge under the current thresholds, next sample that moves the thresholds (tfa2.cpp:363-369)
			const unsigned long long m1 = __ballot(valid && ld > hi);
			const unsigned long long m0 = __ballot(valid && ld < lo) & ~m1;
			const unsigned long long cand = (last_bit ? m0 : m1) & rest;
			const int kc = cand ? __builtin_ctzll(cand) : 64;
			int ku = 64;
			if (bitcnt < 10) {
				const unsigned long long u = __ballot(valid && (ld > dmax || ld < dmin)) & rest;
				ku = u ? __builtin_ctzll(u) : 64;
			}
			const int ke = kc < ku ? kc : ku;
			const int kend = ke < 64 ? ke : nv - 1;  // the stretch [pos, kend] has constant thresholds and bitcnt
			if (bitcnt > 4 && bitcnt < 10) {  // tfa2.cpp:371-375, sample by sample (wrapping int32)
				for (int k = pos; k <= kend; k++) {
					const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pw, k);
					const uint32_t t = (uint32_t)rssi_i + pk;
					rssi_i = (int)((uint32_t)rssi_i + (uint32_t)((int)t / 100));
				}
			}
			if (ke >= 64)
				break;
			if (ku <= kc) {  // the sample moves dmax / dmin; its own edge test uses the new thresholds
				const int ldk = __builtin_amdgcn_readlane(ld, ku);
				if (ldk > dmax)
					dmax = (7 * dmax + ldk) / 8;
				if (ldk < dmin)
					dmin = (7 * dmin + ldk) / 8;
				offset = (dmax + dmin) / 2;
				thresholds();
				const int bitk = ldk > hi ? 1 : 0;
				if ((ldk > hi || ldk < lo) && bitk != last_bit)
					candidate(gb + ku, bitk);
			} else {
				candidate(gb + kc, last_bit ^ 1);
			}
			pos = ke + 1;
		}
	}
	}
	};
	// ---- 64 steps (4096 samples) at a time with a STEP PER LANE, once the thresholds are frozen (round 5).  A lane walks the
	// candidates of its own step exactly as walk_one_block does -- in ABSOLUTE index units, where demodulator::start's rebase
	// (decoder.cpp:118-122) is the identity unless last_bit_idx is block-relative 0 when a block begins: a value set at a
	// block's first sample and still standing 8192 samples later.  A group is shorter than a block, so that can only be the
	// value a group is ENTERED with (then it is left to the scalar walk); a value set at a block's first sample inside the
	// group leaves it as the relative 0 it is -- from a start state (last_bit, last_bit_idx) that is first SPECULATED: last_bit = the
	// polarity of the nearest sample beyond a threshold before the lane, last_bit_idx = the nearest alternation of polarity
	// before it -- what the state is if every edge before the lane was accepted (96 % of the edges are).  Then every lane's
	// start state is compared with what the lane before it really left behind; the lanes that were wrong get the true
	// value and walk again, until nothing changes (lane 0 starts from the true state, so by induction every lane then did;
	// more than 16 rounds, more than 64 bits in a lane: the group is left to the scalar walk).  ~2 walks of ~200 vector
	// instructions per 64 steps instead of 64 x (51 scalar instructions per accepted edge + the step's own ~25).
	auto group_vec = [&](const int gs) -> bool {
		int lb = lbi, cb = cur_block;
		if ((gs >> 13) != cb) {
			lb = rebase_lbi(lb, cb, gs >> 13);
			cb = gs >> 13;
		}
#ifdef TFREC_AMD_VECSTAT
		if (lane == 0) {
			atomicAdd(&T.stats[12], 1ull);
			if (lb == 0)
				atomicAdd(&T.stats[13], 1ull);
		}
#endif
		if (lb == 0)
			return false;  // (block-relative 0 is the reference's "no rebase" value)
		const int Labs = lb + kIndexSpan * cb;
		const int ng = ((last - gs) >> 6) + 1 < 64 ? ((last - gs) >> 6) + 1 : 64;
		const int gl = gs + 64 * lane;
		const int Ibase = 2 * gl;
		// ---- the step's samples against the thresholds: 64-bit masks, bit k = sample gl + k
		unsigned long long mH = 0ull, mL = 0ull;
		if (gl <= last) {
			const uint4 *src = reinterpret_cast<const uint4 *>(ldrow + (gl - og));
			uint4 v[8];
#pragma unroll
			for (int q = 0; q < 8; q++)
				v[q] = gl + 8 * q <= last ? src[q] : make_uint4(0u, 0u, 0u, 0u);
			uint32_t rh[2] = { 0u, 0u }, rl[2] = { 0u, 0u };
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const uint32_t d4[4] = { v[q].x, v[q].y, v[q].z, v[q].w };
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const int s0 = (int)(int16_t)(d4[e] & 0xffffu), s1 = (int)d4[e] >> 16;
					// (bits shifted in by an add-with-carry: the word comes out bit-reversed)
					asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rh[q >> 2]) : "v"(s0), "v"(hi) : "vcc");
					asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rl[q >> 2]) : "v"(s0), "v"(lo) : "vcc");
					asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rh[q >> 2]) : "v"(s1), "v"(hi) : "vcc");
					asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rl[q >> 2]) : "v"(s1), "v"(lo) : "vcc");
				}
			}
			mH = (unsigned long long)__builtin_bitreverse32(rh[0]) | ((unsigned long long)__builtin_bitreverse32(rh[1]) << 32);
			mL = (unsigned long long)__builtin_bitreverse32(rl[0]) | ((unsigned long long)__builtin_bitreverse32(rl[1]) << 32);
			const int nv = last - gl + 1;
			if (nv < 64) {
				const unsigned long long vm = (1ull << nv) - 1ull;
				mH &= vm;
				mL &= vm;
			}
			mL &= ~mH;
		}
		// ---- speculated start states
		const unsigned long long anym = mH | mL;
		int sb_, sl_;
		{
			const int th = mH ? 63 - (int)__builtin_clzll(mH) : -1, tl = mL ? 63 - (int)__builtin_clzll(mL) : -1;
			const unsigned long long gen = __ballot(anym != 0ull && th > tl), prop = __ballot(anym == 0ull);
			const unsigned long long cinm = ((gen | prop) + gen + (unsigned long long)last_bit) ^ prop;
			sb_ = (int)((cinm >> lane) & 1ull);
			// the lane's alternations if every one is accepted: the polarity before every bit is a carry chain
			const unsigned long long sum = (mH | ~anym) + mH + (unsigned long long)sb_;
			const unsigned long long before = sum ^ ~anym;
			const unsigned long long edges = (mH & ~before) | (mL & before);
			const int myedge = Ibase + 2 * (63 - (int)__builtin_clzll(edges | 1ull));
			const unsigned long long he = __ballot(edges != 0ull);
			const unsigned long long below = he & ((1ull << lane) - 1ull);
			const int from = below ? 63 - (int)__builtin_clzll(below) : 0;
			const int got = __shfl(myedge, from, 64);
			sl_ = below ? got : Labs;
		}
		// ---- walk, compare, walk again
		unsigned long long acc = 0ull;
		int cnt = 0, bc = 0, lb_out = sb_, l_out = sl_;
		bool bad = false, dirty = true;
		const int acc_lo = td_lo > 9 ? td_lo : 9;
		const uint32_t acc_span = (uint32_t)(td_hi - acc_lo);
		int rounds = 0;
		while (true) {
			if (dirty) {
				acc = 0ull;
				cnt = 0;
				bc = 0;
				bad = false;
				int lbv = sb_, lv = sl_;
				unsigned long long todo = ~0ull;
				while (true) {
					const unsigned long long m = (lbv ? mL : mH) & todo;
					if (!m)
						break;
					const int k = __builtin_ctzll(m);
					todo = ~1ull << k;
					const int index = Ibase + 2 * k, d = index - lv;
					lv = d > 2 ? index : lv;  // tfa2.cpp:410-411 (d was taken first: the edge's timing uses the old value)
					bc += d > 8 ? 1 : 0;  // tfa2.cpp:391-392
					if ((uint32_t)(d - acc_lo) <= acc_span) {
						const int numbits = tfa2_numbits_mul(d, nb_mul);
						const int run = (numbits < 32 && numbits > 1) ? numbits - 1 : 0;
						if (cnt + run + 1 > 64) {
							bad = true;
						} else {
							acc |= (unsigned long long)((1u << run) - (uint32_t)lbv) << cnt;
							cnt += run + 1;
						}
						lbv ^= 1;
						continue;
					}
					// not accepted: the run of candidates of the same polarity right behind it only moves last_bit_idx
					const unsigned long long rest = m >> 1 >> k;
					const int R = __builtin_ctzll(~rest);
					if (R > 0) {
						const int e = index + 2 - lv;
						const int t_set = e > 2 ? 1 : ((2 - e) >> 1) + 2;
						if (t_set <= R)
							lv = index + 2 * (t_set + 2 * ((R - t_set) >> 1));
						todo = ~1ull << (k + R);
					}
				}
				lb_out = lbv;
				l_out = lv;
			}
			// what the lane before left behind (lane 0: the state the group was entered with)
			int pb = __shfl_up(lb_out, 1, 64), pl = __shfl_up(l_out, 1, 64);
			if (lane == 0) {
				pb = last_bit;
				pl = Labs;
			}
			dirty = pb != sb_ || pl != sl_;
			sb_ = pb;
			sl_ = pl;
			if (__ballot(dirty) == 0ull)
				break;
			if (++rounds > 16) {
#ifdef TFREC_AMD_VECSTAT
				if (lane == 0)
					atomicAdd(&T.stats[14], 1ull);
#endif
				return false;
			}
		}
#ifdef TFREC_AMD_VECSTAT
		{
			const bool anybad = __ballot(bad) != 0ull;
			if (lane == 0) {
				atomicAdd(&T.stats[6], (unsigned long long)(rounds + 1));
				atomicAdd(&T.stats[15], anybad ? 1ull : 0ull);
			}
		}
#endif
		if (__ballot(bad) != 0ull)
			return false;
		coop_join_bits(bw, stage, acc, cnt);
		// ---- commit the group
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1)
			bc += __shfl_xor(bc, o, 64);
		bitcnt += bc;
		last_bit = __builtin_amdgcn_readlane(lb_out, 63);
		const int Lnew = __builtin_amdgcn_readlane(l_out, 63);
		const int gend = gs + 64 * ng - 1 < last ? gs + 64 * ng - 1 : last;
		cur_block = gend >> 13;
		lbi = Lnew - kIndexSpan * cur_block;
		return true;
	};
	if (!(stage && T.tfa2_vec && nb_mul)) {
		old_range(g1, last + 1);
	} else {
		for (int pos = g1; pos <= last;) {
			if (bitcnt >= 10) {
				if (!group_vec(pos)) {
					gs.scalar++;  // a group left to the scalar walk (tfrec_amd_get_stats)
					old_range(pos, pos + 4096);
				} else {
					gs.vector++;
				}
				pos += 4096;
			} else {  // the thresholds still adapt (a head that gave up): stretch by stretch
				old_range(pos, pos + 256);
				pos += 256;
			}
		}
	}
	const int bl = last >> 13;
	if (bl != cur_block) {
		lbi = rebase_lbi(lbi, cur_block, bl);
		cur_block = bl;
	}
	if (closed)  // tfa2.cpp:430-431: trailing bits before the flush
		bw.put_run(last_bit, 16);
	bw.finish();
#ifdef TFREC_AMD_COOPSTAT
	if (lane == 0) {
		atomicAdd(&T.stats[7], cs_steps);
		atomicAdd(&T.stats[8], cs_acc);
		atomicAdd(&T.stats[9], cs_rej);
		atomicAdd(&T.stats[10], cs_slow);
		atomicAdd(&T.stats[13], cs_full);
		atomicAdd(&T.stats[14], cs_pop);
		atomicAdd(&T.stats[15], cs_cont);
	}
#endif
	if (lane == 0) {
		WinResult r;
		r.nbits = bw.n;
		r.closed = closed ? 1 : 0;
		r.rssi_i = rssi_i;
		r.offset = offset;
		r.lbi_out = lbi;
		r.first_cand_g = first_cand_g;
		r.bitcnt = bitcnt;
		r.dmin = dmin;
		r.dmax = dmax;
		r.last_bit = last_bit;
		r.mark_lvl = 0;
		r.resume = -1;
		rr = r;
	}
}

// TFA_1, 64 steps (4096 samples) at a time with a STEP PER LANE (round 5).  The scalar walk further down spends ~55 scalar
// instructions on every run of candidates and ~60 on every step (2.0 M runs in 2.2 M steps per benchmark batch: the most
// expensive code of the batch after the TFA_2 walk).  What makes the lane-parallel form exact:
//   * In ABSOLUTE index units I = 2 * (sample of the submit) demodulator::start's rebase (decoder.cpp:118-122) is the identity
//     for every value but a block-relative 0, which can only come about when a candidate at a block's first sample sets it
//     (tfa1.cpp:175-176 with index 0) and which the demodulator reads as "no pulse yet" (:165).  So: an absolute value at a
//     block's first sample (a multiple of 16384) means "none" -- the next run's first sample emits nothing -- and everything
//     else is plain arithmetic.  (Within the run that set it the relative 0 is also the true relative index: the closed
//     forms hold.)  Only a group that is ENTERED with "none" and has a candidate at a block's second sample (index 2:
//     "index - 0 > 2" does not fire, the value stays "none") is left to the scalar walk.
//   * A maximal run of candidates that begins at I0 behind a non-candidate finds I0 - lbi >= 4: the rule "index - lbi > 2"
//     sets lbi = I0 whatever lbi was, so what the run leaves behind (I0 + 4 * ((len - 1) >> 1)) does not depend on history;
//     only the bits its FIRST sample emits do (the gap to what the run before it left behind: tfa1.cpp:167-173).
//   * A run that crosses a step boundary continues in the next lane with I0 - lbi = 2 or 4, 2 iff the run has had an odd number
//     of samples so far (by the same closed form the scalar walk uses for the rest of a run): a parity, generated by every
//     lane whose word ends in an odd number of ones, handed through words that are all ones -- the carries of ONE 64-bit
//     addition of two ballots.
// So: every lane walks the runs of its own 64-bit candidate word with the scalar walk's formulas (a lane's first run
// either continues the lane before it, or it is a maximal run's beginning and only its emission waits for the value the
// nearest lane with candidates before it leaves behind); the lanes' bits (at most 64 each, else the group is left to the
// scalar walk) are joined by a prefix sum through an LDS image of the output words.  ~400 instructions per 64 steps
// instead of ~7000.  mark_kernel's pieces (16 steps each) are checked for the whole group first.
__device__ __forceinline__ void coop_tfa1(int c, int j, int n_streams, int M, const uint32_t *__restrict__ dec,
					  size_t dec_stride, const ChainLaunch &L, const WinTables &T, int *__restrict__ lds_m,
					  uint32_t *__restrict__ stage, GroupStats &gs)
{
	const int lane = threadIdx.x;
	const int a = c / n_streams, s = c - a * n_streams;
	const ChainState &st = L.states[a][s];
	const int og = T.open[(size_t)c * T.cap + j];
	const int close = T.close[(size_t)c * T.cap + j];
	const bool closed = close < M;
	const int last = closed ? close : M - 1;
	const bool cont = (j == 0) && T.cont[c];
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const uint32_t prev0 = ((uint32_t)st.prev_i & 0xffffu) | ((uint32_t)st.prev_q << 16);
	int mark = 0, lbi = 0;  // tfa1.cpp:183: the window opens with last_bit_idx = 0
	int cur_block = og >> 13;
	int rssi_lane = 0;
	if (cont) {  // resume the window the previous submit left open
		mark = st.mark_lvl;
		rssi_lane = st.rssi_i;
		lbi = rebase_lbi(st.last_bit_idx, -1, cur_block);
	}
	CoopBits bw;
	bw.init(T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j, 0);
	auto load = [&](int g) -> uint2 {  // (sample, previous sample)
		uint2 v;
		v.x = drow[g];
		v.y = g > 0 ? drow[g - 1] : prev0;
		return v;
	};
	// mark_kernel has run the peak detector of every 1024-sample piece from a warm-up: its result is used when the
	// value it started the piece from is the true one, otherwise the piece is recomputed here (wave-uniform)
	const int slot0 = win_slot0(og, j);
	const uint32_t *candrow = T.cand + (size_t)s * T.slots + slot0;
	const MarkPiece *markrow = T.mark + (size_t)s * T.slots + slot0;
	const int nsteps = ((last - og) >> 6) + 1;
	const bool use_vec = T.tfa1_vec != 0;
	for (int sb = 0; sb < nsteps; sb += 64) {
		// the candidate words of 64 steps at a time, a step per lane (fetched per step they were two scalar loads the wave
		// waited for in every step); bits behind the window's last sample are zero (mark_kernel), a half it did not write is not read
		const int sl = sb + lane;
		const int gb_l = og + 64 * sl;
		const uint32_t cw_lo = gb_l <= last ? candrow[2 * sl] : 0u;
		const uint32_t cw_hi = gb_l + 32 <= last ? candrow[2 * sl + 1] : 0u;
		const int ng = nsteps - sb < 64 ? nsteps - sb : 64;
		bool done = false;
		if (use_vec) {
			// ---- the group's pieces: each must have started from the true value
			const int np = (ng + kMarkSlots / 2 - 1) / (kMarkSlots / 2);
			MarkPiece mpl = { 0, 0, 0, 0 };
			if (lane < np)
				mpl = markrow[2 * (sb + (kMarkSlots / 2) * lane)];
			int mk = mark, rmax = 0;
			bool ok = true;
#pragma unroll
			for (int p = 0; p < 4; p++) {
				if (p < np) {
					ok = ok && __builtin_amdgcn_readlane(mpl.start, p) == mk;
					mk = __builtin_amdgcn_readlane(mpl.end, p);
					const int mx_ = __builtin_amdgcn_readlane(mpl.max, p);
					rmax = mx_ > rmax ? mx_ : rmax;
				}
			}
			const unsigned long long w = (unsigned long long)cw_lo | ((unsigned long long)cw_hi << 32);
			const bool have = lbi != 0;
			const int Labs = have ? lbi + kIndexSpan * cur_block : kIndexSpan * ((og + 64 * sb) >> 13);  // ("none": a block's first sample)
			// ---- entered with "none": a candidate at a block's second sample would leave it standing
			const int rel = gb_l & (kBlockDec - 1);
			const int d1 = (kBlockDec + 1 - rel) & (kBlockDec - 1);
			const bool hz = !have && d1 < 64 && ((w >> d1) & 1ull);
#ifdef TFREC_AMD_VECSTAT
			{
				const bool anyhz = __ballot(hz) != 0ull;
				if (lane == 0) {
					atomicAdd(&T.stats[8], 1ull);
					if (!ok)
						atomicAdd(&T.stats[9], 1ull);
					else if (anyhz)
						atomicAdd(&T.stats[10], 1ull);
				}
			}
#endif
			ok = ok && __ballot(hz) == 0ull;
			if (ok) {
				// ---- does a lane's first run continue the lane before it, and with which parity
				const int Ibase = 2 * gb_l;
				const int g0 = 2 * (og + 64 * sb) - Labs;  // the group's first sample against last_bit_idx
				const bool cont0 = have && g0 <= 4 && (__builtin_amdgcn_readlane((int)cw_lo, 0) & 1);
				const int q = ~w ? (int)__builtin_clzll(~w) : 64;  // ones at the word's top
				const unsigned long long pm = __ballot(q < 64 && (q & 1)), fm = __ballot(q == 64);
				const unsigned long long cin = (((pm | fm) + pm + ((cont0 && g0 == 2) ? 1ull : 0ull)) ^ fm);
				const int up = __shfl_up((int)(cw_hi >> 31), 1, 64);
				const bool cont_l = lane == 0 ? cont0 : ((cw_lo & 1u) && up);
				int Lc = Ibase - (((cin >> lane) & 1ull) ? 2 : 4);  // (a continued run's last_bit_idx; else set below)
				unsigned long long ww = w, acc = 0ull;
				int cnt = 0, I0f = 0;
				bool bad = false, defer = false, first = true;
				while (__ballot(ww != 0ull) != 0ull) {
					if (ww != 0ull) {
						const int k0 = __builtin_ctzll(ww);
						const unsigned long long inv = ~(ww >> k0);
						const int len = inv ? __builtin_ctzll(inv) : 64 - k0;  // run of consecutive candidates
						ww = (k0 + len >= 64) ? 0ull : (ww & (~0ull << (k0 + len)));
						const int I0 = Ibase + 2 * k0;
						if (first && !cont_l) {  // a maximal run begins: gap >= 4, lbi = I0; its bits wait for the gap
							defer = true;
							I0f = I0;
							Lc = I0;
						} else {  // first sample of the run: tfa1.cpp:165-177
							const int gap = I0 - Lc;
							if (gap > 4 && (Lc & (kIndexSpan - 1)) != 0) {  // (tfa1.cpp:165: a block-relative 0 is "no pulse yet")
								const int ones = gap >= 22 ? (gap - 22) / 20 + 1 : 0;  // ones for n = 22, 42, ... <= gap
								if (ones >= 32 || cnt + ones + 1 > 64) {
									bad = true;
								} else {
									acc |= ((1ull << ones) - 1ull) << cnt;  // ... and the zero behind them
									cnt += ones + 1;
								}
							}
							if (gap > 2)
								Lc = I0;
						}
						first = false;
						// the rest of the run: every gap is <= 4, so nothing is emitted; last_bit_idx follows "index - lbi > 2"
						if (len > 1) {
							const int d = I0 - Lc;          // 0 (just set) or 2
							const int t1 = d >= 2 ? 1 : 2;  // first t >= 1 with I0 + 2t - lbi > 2
							if (t1 <= len - 1)
								Lc = I0 + 2 * t1 + 4 * ((len - 1 - t1) >> 1);
						}
					}
				}
				// ---- the deferred first runs: the gap to what the nearest lane with candidates before leaves behind
				const unsigned long long ne = __ballot(w != 0ull);
				const unsigned long long below = ne & ((1ull << lane) - 1ull);
				const int src = below ? 63 - (int)__builtin_clzll(below) : 0;
				const int Lsrc = __shfl(Lc, src, 64);
				const int Lprev = below ? Lsrc : Labs;
				// (A lane's LATER runs lie within 64 samples of the one before: at most 6 ones.  Its first run can come after any
				// silence -- another protocol's burst holds the deviation up for thousands of samples --: 32 ones or more go through
				// the wave-uniform writer, between the lanes before and this lane's other bits.)
				int ones_long = 0;
				if (defer && (Lprev & (kIndexSpan - 1)) != 0) {
					const int gap = I0f - Lprev;
					if (gap <= 2)
						bad = true;  // (cannot happen: see above)
					if (gap > 4) {
						const int ones = gap >= 22 ? (gap - 22) / 20 + 1 : 0;
						if (ones >= 32) {
							ones_long = ones;
						} else if (cnt + ones + 1 > 64) {
							bad = true;
						} else {
							acc = (acc << (ones + 1)) | ((1ull << ones) - 1ull);  // they come before the lane's other bits
							cnt += ones + 1;
						}
					}
				}
#ifdef TFREC_AMD_VECSTAT
				{
					const bool anybad = __ballot(bad) != 0ull;
					const unsigned long long nlong = (unsigned long long)__builtin_popcountll(__ballot(ones_long != 0));
					if (lane == 0) {
						atomicAdd(&T.stats[11], anybad ? 1ull : 0ull);
						atomicAdd(&T.stats[5], nlong);
					}
				}
#endif
				if (__ballot(bad) == 0ull) {
					unsigned long long longs = __ballot(ones_long != 0);
					for (int from = 0;;) {
						const int to = longs ? (int)__builtin_ctzll(longs) : 64;
						const bool mine = lane >= from && lane < to;
						coop_join_bits(bw, stage, mine ? acc : 0ull, mine ? cnt : 0);
						if (to == 64)
							break;
						bw.put_run(1, __builtin_amdgcn_readlane(ones_long, to));
						bw.put_run(0, 1);
						longs &= longs - 1ull;
						from = to;
					}
					// ---- commit the group
					mark = mk;
					rssi_lane = rmax > rssi_lane ? rmax : rssi_lane;
					const int gend = og + 64 * (sb + ng) - 1 < last ? og + 64 * (sb + ng) - 1 : last;
					const int nb = gend >> 13;
					const int Lnew = ne != 0ull ? __builtin_amdgcn_readlane(Lc, 63 - (int)__builtin_clzll(ne)) : Labs;
					lbi = (Lnew & (kIndexSpan - 1)) != 0 ? Lnew - kIndexSpan * nb : 0;
					cur_block = nb;
					done = true;
				}
			}
		}
		if (done) {
			gs.vector++;
			continue;
		}
		if (use_vec)
			gs.scalar++;  // a group left to the scalar walk (tfrec_amd_get_stats)
		bool piece_ok = false;
		MarkPiece mp = { 0, 0, 0, 0 };
		for (int step = sb; step < sb + ng; step++) {
			const int gb = og + 64 * step;
			if ((step & (kMarkSlots / 2 - 1)) == 0) {
				mp = markrow[2 * step];
				piece_ok = mp.start == mark;
				if (piece_ok && mp.max > rssi_lane)
					rssi_lane = mp.max;  // tfa1.cpp:161-162
			}
			const int nv = last - gb + 1 < 64 ? last - gb + 1 : 64;
			unsigned long long m;
			if (piece_ok) {
				const unsigned long long lo = (uint32_t)__builtin_amdgcn_readlane((int)cw_lo, step & 63);
				const unsigned long long hi = (uint32_t)__builtin_amdgcn_readlane((int)cw_hi, step & 63);
				m = lo | (hi << 32);
				if (gb + 64 > last || ((step + 1) & (kMarkSlots / 2 - 1)) == 0)
					mark = mp.end;  // the piece ends with this step
			} else {
				const uint2 cur = load(gb + lane <= last ? gb + lane : last);
				const int dev = fm_dev_nrzs((int)(int16_t)(cur.x & 0xffff), (int)cur.x >> 16, (int)(int16_t)(cur.y & 0xffff),
							    (int)cur.y >> 16);
				// the peak detector, wave-uniform (tfa1.cpp:157-160); mark >= 0 always, so (int) truncation is exact
				for (int k = 0; k < nv; k++) {
					const int dk = __builtin_amdgcn_readlane(dev, k);
					mark = dk > mark ? dk : tfa1_decay(mark);
					lds_m[k] = mark;
				}
				__syncthreads();
				const int mk = lds_m[lane];
				__syncthreads();
				const bool valid = lane < nv;
				if (valid && mk > rssi_lane)
					rssi_lane = mk;  // tfa1.cpp:161-162
				m = __ballot(valid && dev < mk / 2);  // tfa1.cpp:164
				atomicAdd(&T.stats[4], lane == 0 ? 1ull : 0ull);  // steps recomputed (tfrec_amd_get_stats)
			}
#ifdef TFREC_AMD_COOPSTAT
			if (lane == 0)
				atomicAdd(&T.stats[11], 1ull);
#endif
			while (m) {
#ifdef TFREC_AMD_COOPSTAT
				if (lane == 0)
					atomicAdd(&T.stats[12], 1ull);
#endif
				const int k0 = __builtin_ctzll(m);
				const unsigned long long inv = ~(m >> k0);
				int len = inv ? __builtin_ctzll(inv) : 64 - k0;  // run of consecutive candidates
				const int g0 = gb + k0;
				const int left_in_block = kBlockDec - (g0 & (kBlockDec - 1));
				if (len > left_in_block)
					len = left_in_block;  // last_bit_idx is rebased at every block start: cut the run there
				m = (k0 + len >= 64) ? 0ull : (m & (~0ull << (k0 + len)));
				const int b = g0 >> 13;
				if (b != cur_block) {
					lbi = rebase_lbi(lbi, cur_block, b);
					cur_block = b;
				}
				const int i0 = 2 * (g0 & (kBlockDec - 1));
				// first sample of the run: tfa1.cpp:165-177
				if (lbi) {
					const int gap = i0 - lbi;
					if (gap > 4) {
						const int ones = gap >= 22 ? (gap - 22) / 20 + 1 : 0;  // ones for n = 22, 42, ... <= gap
						if (ones < 32) {
							bw.put_bits((1u << ones) - 1u, ones + 1);  // ... and the zero behind them, in one go
						} else {
							bw.put_run(1, ones);
							bw.put_run(0, 1);
						}
					}
				}
				if (i0 - lbi > 2)
					lbi = i0;
				// the rest of the run: every gap is <= 4, so nothing is emitted; last_bit_idx follows "index - lbi > 2"
				if (len > 1) {
					const int d = i0 - lbi;               // 0 (just set) or 2
					const int t1 = d >= 2 ? 1 : 2;        // first t >= 1 with i0 + 2t - lbi > 2
					if (t1 <= len - 1)
						lbi = i0 + 2 * t1 + 4 * ((len - 1 - t1) >> 1);
				}
			}
		}
	}
	const int bl = last >> 13;
	if (bl != cur_block) {
		lbi = rebase_lbi(lbi, cur_block, bl);
		cur_block = bl;
	}
	bw.finish();
	// rssi = max over the lanes
	int rssi = rssi_lane;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const int v = __shfl_xor(rssi, o, 64);
		rssi = v > rssi ? v : rssi;
	}
	if (lane == 0) {
		WinResult r;
		r.nbits = bw.n;
		r.closed = closed ? 1 : 0;
		r.rssi_i = rssi;
		r.offset = 0;
		r.lbi_out = lbi;
		r.first_cand_g = -1;
		r.bitcnt = 0;
		r.dmin = 32767;
		r.dmax = -32767;
		r.last_bit = 0;
		r.mark_lvl = mark;
		r.resume = -1;
		T.result[(size_t)c * T.cap + j] = r;
	}
}

__global__ __launch_bounds__(64) void coop_slicer_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							 const int16_t *__restrict__ ld16, int n_streams, int n_blocks,
							 ChainLaunch L, WinTables T, int kind)
{
	__shared__ int lds_m[64];
	__shared__ uint32_t t1_stage[kCoopStageWords];
	latency_prio();
	const int M = n_blocks * kBlockDec;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	const int q = 2 * kind;  // the long windows of this kind
	const uint32_t count = T.queue[q].count;
	GroupStats gs = { 0, 0 };
	for (uint32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {  // wave-uniform
		const uint2 it = T.items[(size_t)q * total + idx];
		const int c = __builtin_amdgcn_readfirstlane((int)it.x), j = __builtin_amdgcn_readfirstlane((int)it.y);
		if (kind == 0)
			coop_tfa1(c, j, n_streams, M, dec, dec_stride, L, T, lds_m, t1_stage, gs);
		else
			coop_tfa2(c, j, n_streams, M, dec, dec_stride, ld16, L, T, t1_stage, gs);
	}
	stat_flush(T, gs, kind == 0 ? kStatTfa1Scalar : kStatTfa2Scalar, kind == 0 ? kStatTfa1Vector : kStatTfa2Vector);
}

constexpr int kWhbRunEsc = 0xffff;            // run-length escape: the next two uint16 hold a 32-bit length

// ------------------------------------------------------------------------------------------------ K4'' WHB commit
// whb_decoder::store_bit (whb.cpp:566-603) over the runs whb_demod_kernel accepted, in two stages like K5:
//   whb_decode_window  lane per WINDOW, from the decoder registers whb_demod_kernel recorded at the window's first
//                      bit: replays the runs, collects the rdata bytes the window writes (a 64-bit written-mask:
//                      before a stream's first flush bytes are also stored without a sync word);
//   whb_commit_stream  lane per stream: overlays the windows' bytes in order, reports the flushes (whb.cpp:693-697),
//                      commits the decoder state.
// Both run in the tail of whb_demod_kernel, by the wave that demodulated the stream (they were kernels of their own:
// two more launches on the longest chain of the batch, each waiting its turn for the chip).
__device__ __forceinline__ void whb_store_bit_m(Dec &d, int bit, unsigned long long &wmask)
{
	if (bit == d.w_last_bit)
		d.psk = 1 - d.psk;
	if (d.psk == d.last_psk)
		d.nrzs = 1 - d.nrzs;
	d.w_last_bit = bit;
	d.last_psk = d.psk;
	const int out = d.nrzs ^ ((d.lfsr >> 16) & 1) ^ ((d.lfsr >> 11) & 1);
	d.lfsr = (d.lfsr << 1) | (uint32_t)d.nrzs;
	d.sr = (d.sr >> 1) | ((uint32_t)out << 31);
	if (d.sr == 0x2bd42d4bu) {
		d.synced = 1;
		d.sr_cnt = 0;
		d.rdata[0] = d.sr & 0xff;
		d.rdata[1] = (d.sr >> 8) & 0xff;
		d.rdata[2] = (d.sr >> 16) & 0xff;
		d.byte_cnt = 3;
		wmask |= 7ull;
	}
	if (d.sr_cnt == 0) {
		if (d.byte_cnt < 64) {  // only rdata[0 .. 64) is ever looked at (flush reads r[plen + 3], plen <= 60: whb.cpp:484-510)
			d.rdata[d.byte_cnt] = (d.sr >> 24) & 0xff;
			wmask |= 1ull << d.byte_cnt;
		}
		d.byte_cnt++;
	}
	if (d.sr_cnt >= 0)
		d.sr_cnt = (d.sr_cnt + 1) & 7;
}

// one lane: window j of stream s
__device__ __forceinline__ void whb_decode_window(int s, int j, int n_streams, const ChainLaunch &L, int a, const WinTables &T,
						  uint8_t *__restrict__ my_rdata)
{
	{
		const int c = a * n_streams + s;
		const ChainState &st = L.states[a][s];
		const WinResult r = T.result[(size_t)c * T.cap + j];
		const int og = T.open[(size_t)c * T.cap + j];
		const uint32_t *ent32 = T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j;
		unsigned long long wmask = 0;
		Dec d{ 0u, -1, 0, 0, 0, 0, 0, 0, 0, 0u, 0u, my_rdata };
		if (j == 0) {  // continues from the carried decoder state
			const uint4 *src = reinterpret_cast<const uint4 *>(st.rdata);
			uint4 *dst = reinterpret_cast<uint4 *>(my_rdata);
#pragma unroll
			for (int q = 0; q < 4; q++)
				dst[q] = src[q];
			d.sr = st.sr;
			d.sr_cnt = st.sr_cnt;
			d.byte_cnt = st.byte_cnt;
			d.synced = st.synced;
			d.w_last_bit = st.w_last_bit;
			d.nrzs = st.nrzs;
			d.lfsr = st.lfsr;
			wmask = ~0ull;
		} else {
			const WhbStart ws = T.whbstart[(size_t)s * T.cap + j];
			d.sr = ws.sr;
			d.sr_cnt = ws.sr_cnt;
			d.byte_cnt = ws.byte_cnt;
			d.synced = ws.synced;
			d.lfsr = ws.lfsr;
			d.nrzs = (int)(ws.lfsr & 1u);
			d.w_last_bit = d.nrzs ^ ((st.nrzs ^ st.w_last_bit) & 1);  // nrzs(t) = bit(t) ^ K, K fixed per stream
		}
		// psk is tracked relative to 0 (store_bit always leaves last_psk == psk; only its parity is carried on)
		const int nent = r.nbits;
		int q = 0, widx = -1;
		uint32_t wcur = 0, wnext = nent > 0 ? ent32[0] : 0u;
		while (q < nent) {
			const int wi = q >> 1;
			if (wi != widx) {
				wcur = wi == widx + 1 ? wnext : ent32[wi];
				widx = wi;
				if (2 * (wi + 1) < nent)
					wnext = ent32[wi + 1];  // in flight while this word's runs are decoded
			}
			int len = (q & 1) ? (int)(wcur >> 16) : (int)(wcur & 0xffff);
			q++;
			if (len == kWhbRunEsc) {
				const uint16_t *e16 = reinterpret_cast<const uint16_t *>(ent32);
				len = (int)((uint32_t)e16[q] | ((uint32_t)e16[q + 1] << 16));
				q += 2;
			}
			whb_store_bit_m(d, 0, wmask);  // whb.cpp:666-673: one 0, then (len - 1) ones
			for (int m = 1; m < len; m++)
				whb_store_bit_m(d, 1, wmask);
		}
		if (r.closed)  // the window ends with a flush (whb.cpp:693-697): 16 x store_bit(0) first
			for (int z = 0; z < 16; z++)
				whb_store_bit_m(d, 0, wmask);
		WinDecode &o = T.decode[(size_t)c * T.cap + j];
		o.sr = d.sr;
		o.sr_cnt = d.sr_cnt;
		o.byte_cnt = d.byte_cnt;
		o.invert = (d.psk ? kWhbFPsk : 0) | (d.synced ? kWhbFSynced : 0) | (d.w_last_bit ? kWhbFLastBit : 0) |
			   (d.nrzs ? kWhbFNrzs : 0);
		o.wlen = 0;
		o.lfsr = d.lfsr;
		o.wmask = wmask;
		const uint4 *src = reinterpret_cast<const uint4 *>(my_rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(o.vals);
#pragma unroll
		for (int q4 = 0; q4 < 4; q4++)
			dst[q4] = src[q4];
	}
}

// one lane: stream s
__device__ __forceinline__ void whb_commit_stream(int s, int n_streams, int n_blocks, long long sample_base, const ChainLaunch &L,
						  int a, const WinTables &T, tfrec_amd_event *__restrict__ events,
						  EventBuf *__restrict__ eb, uint32_t flags, uint8_t *__restrict__ my_rdata)
{
	const int M = n_blocks * kBlockDec;
	const ChainParams &p = L.params[a];
	ChainState &st = L.states[a][s];
	const int c = a * n_streams + s;
	const int count = T.count[c];
	EmitCtx e{ events, eb, flags, (uint32_t)s, L.slot[a], p.sensor_type, sample_base };
	{  // rdata[0 .. 64) as the previous submit left them (only these are ever looked at: INTEGRATION.md)
		const uint4 *src = reinterpret_cast<const uint4 *>(st.rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(my_rdata);
#pragma unroll
		for (int q = 0; q < 4; q++)
			dst[q] = src[q];
	}
	Dec d{ st.sr, st.sr_cnt, st.byte_cnt, st.invert, st.synced, st.w_last_bit, st.psk, st.last_psk, st.nrzs, st.lfsr, st.seq,
	       my_rdata };
	for (int j = 0; j < count; j++) {
		const int close = T.close[(size_t)c * T.cap + j];
		const int last = close < M ? close : M - 1;
		const WinResult *rr = &T.result[(size_t)c * T.cap + j];
		const WinDecode *wd = &T.decode[(size_t)c * T.cap + j];
		// the window's rdata writes on top of what was there
		const unsigned long long wm = wd->wmask;
		const uint32_t *vsrc = reinterpret_cast<const uint32_t *>(wd->vals);
		uint32_t *vdst = reinterpret_cast<uint32_t *>(my_rdata);
		if (wm)
			for (int w = 0; w < 16; w++) {
				const uint32_t nib = (uint32_t)(wm >> (4 * w)) & 15u;
				const uint32_t m = ((nib & 1u) ? 0xffu : 0u) | ((nib & 2u) ? 0xff00u : 0u) | ((nib & 4u) ? 0xff0000u : 0u) |
						   ((nib & 8u) ? 0xff000000u : 0u);
				vdst[w] = (vdst[w] & ~m) | (vsrc[w] & m);
			}
		const int fl = wd->invert;
		d.sr = wd->sr;
		d.sr_cnt = wd->sr_cnt;
		d.byte_cnt = wd->byte_cnt;
		d.synced = (fl & kWhbFSynced) ? 1 : 0;
		d.w_last_bit = (fl & kWhbFLastBit) ? 1 : 0;
		d.nrzs = (fl & kWhbFNrzs) ? 1 : 0;
		d.psk ^= (fl & kWhbFPsk) ? 1 : 0;
		d.last_psk = d.psk;
		d.lfsr = wd->lfsr;
		if (flags & TFREC_AMD_F_BITS) {  // parity mode: the runs "0,1,1,.." (and the 16 zeros before a flush) as bits
			const int og = T.open[(size_t)c * T.cap + j];
			const uint16_t *e16 = reinterpret_cast<const uint16_t *>(T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j);
			uint32_t words[16];
			int nb = 0, chunk = 0;
			auto put = [&](int bit) {
				if ((nb & 31) == 0)
					words[nb >> 5] = 0u;
				words[nb >> 5] |= (uint32_t)bit << (nb & 31);
				if (++nb == 512) {
					emit_bits(e, d.seq, og, chunk, words, 512);
					chunk++;
					nb = 0;
				}
			};
			for (int q = 0; q < rr->nbits;) {
				int len = e16[q++];
				if (len == kWhbRunEsc) {
					len = (int)((uint32_t)e16[q] | ((uint32_t)e16[q + 1] << 16));
					q += 2;
				}
				put(0);
				for (int m = 1; m < len; m++)
					put(1);
			}
			if (rr->closed)
				for (int z = 0; z < 16; z++)
					put(0);
			if (nb)
				emit_bits(e, d.seq, og, chunk, words, nb);
		}
		if (rr->closed) {  // whb.cpp:693-697
			const long long rssi =
				(long long)((unsigned long long)(uint32_t)rr->rssi_i | ((unsigned long long)(uint32_t)rr->offset << 32));
			// (the event's index: should the stream's speculation turn out wrong, the exact kernel retracts the event)
			T.result[(size_t)c * T.cap + j].first_cand_g = flush<2>(e, d, rssi, 0, last);
		}
	}
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(my_rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(st.rdata);
#pragma unroll
		for (int q2 = 0; q2 < 4; q2++)
			dst[q2] = src[q2];
	}
	st.sr = d.sr;
	st.sr_cnt = d.sr_cnt;
	st.byte_cnt = d.byte_cnt;
	st.synced = d.synced;
	st.w_last_bit = d.w_last_bit;
	st.psk = d.psk;
	st.last_psk = d.last_psk;
	st.nrzs = d.nrzs;
	st.lfsr = d.lfsr;
	st.seq = d.seq;
}


// ------------------------------------------------------------------------------------------------ K4' WHB stage 2
// whb_demod::demod after the first low-pass (whb.cpp:653-703): ONE WAVE PER STREAM, 64 samples per step.
//
// The decision-level average (iir_avg, whb.cpp:654) is a non-contracting biquad that only runs while the decoder is
// unsynced -- it can neither be speculated nor separated from the bit decisions, so a stream is one serial chain of
// ~120 k recurrence steps per batch, and with ~1000 streams there is one such wave per SIMD: the kernel's duration is
// the number of instructions ONE wave issues (a lone wave issues one instruction per 4-8 cycles whatever the lane
// count).  Everything here is arranged to keep that count down:
//   * windows are the outer loop, the steps of a window the inner one (contiguous addresses, two loads in flight);
//   * per step, lane n owns sample n: neighbours by DPP wave shifts, the feed-forward terms of the biquad in the
//     3-multiply form of iir_step_t() (b1 = 2 b0, b2 = b0: P = fma(2, t1, t0), B2 = t2, t = fl((b0/2) * dev));
//   * the 64-step feedback recurrence y = ((B2 + a1*y1) + P) + a2*y2 runs on all lanes redundantly, fully unrolled
//     behind register-resident feed-forward pairs (5 fp64 operations + 1.5 LDS instructions per sample: the serial
//     floor); lane n reads y(n) back, "dev < avg_of && dev > last_dev" (whb.cpp:662-663) is one ballot;
//   * the accepted candidates (spacing rule :664; about one per step) emit runs "0,1,1,.." whose lengths are
//     collected lane-per-entry in a register and stored 64 at a time; has_sync() is tracked without a per-bit loop:
//     store_bit leaves last_psk == psk, hence nrzs(t) = bit(t) ^ K and the descrambled bit is
//     nrzs(t) ^ nrzs(t-12) ^ nrzs(t-17) (whb.cpp:568-580) -- GF(2)-linear, so the 32-bit sync compare is evaluated for
//     all positions of a run at once, one position per lane;
//   * once the decoder has locked (until the window's flush) a step is only the candidate test against the frozen
//     average plus a per-lane power sum (whb.cpp:677-678: exact integers, reduced once per window).
// When the decoder locks at sample k of a step, y(0..k) is already in LDS: the filter state is taken at k and the
// candidates after k are re-tested against the frozen average -- no rewind.
// The decoder stages (whb_decode_window, whb_commit_stream) run in the tail, by the same wave.
#ifndef TFREC_AMD_WHB_AHEAD
#define TFREC_AMD_WHB_AHEAD 1
#endif
constexpr int kWhbAhead = TFREC_AMD_WHB_AHEAD;  // whb_demod_kernel: steps whose stage-1 outputs are held ahead of the current one (one more is being loaded)
constexpr int kWhbSpb = 64, kWhbSpbShift = 6;  // whb_demod's samples per bit (main.cpp:217), see whb_demod_kernel
constexpr uint32_t kWhbSyncRev = 0xd2b42bd4u;  // bit-reversed 0x2bd42d4b (whb.cpp:582): newest bit at the LSB

__device__ __forceinline__ int wave_shr1(int v)  // lane n <- lane n-1 (lane 0: 0)
{
	return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}
// out[j] = row j of v in all four rows: lane 16r + i receives v of lane 16j + i (v_permlane16_swap, v_permlane32_swap)
__device__ __forceinline__ void rows_replicate(int v, int (&out)[4])
{
	const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);        // (R0 R0 R2 R2), (R1 R1 R3 R3)
	const auto e = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);  // (R0 x4), (R2 x4)
	const auto o = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);  // (R1 x4), (R3 x4)
	out[0] = e[0];
	out[1] = o[0];
	out[2] = e[1];
	out[3] = o[1];
}
__device__ __forceinline__ void rows_replicate(double v, double (&out)[4])
{
	int lo[4], hi[4];
	rows_replicate(__double2loint(v), lo);
	rows_replicate(__double2hiint(v), hi);
#pragma unroll
	for (int j = 0; j < 4; j++)
		out[j] = __hiloint2double(hi[j], lo[j]);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)  // wave-uniform lane
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// lane n <- lane n - d of its row of 16 (zero for the first d lanes of a row): DPP row_shr with bound_ctrl
template <int D>
__device__ __forceinline__ double row_shr_f64(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + D, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + D, 0xf, 0xf, true);
	return __hiloint2double(hi, lo);
}

// The decision-level average over one 64-sample step, all samples at once (EXACT = false).  y(k) = a1 y(k-1) + a2 y(k-2)
// + x(k) in state form s(k) = M s(k-1) + (x(k), 0), M = [[a1, a2], [1, 0]]: a zero-state scan inside every row of 16
// lanes (four DPP levels with M, M^2, M^4, M^8), the rows' end states E_q by readlane, and the carry-in of the step's
// start state (y1, y2) and of the rows before as per-lane constant rows of powers of M:
//     y(k) = u(k) + R(k+1) . (y1, y2) + sum_{q < k/16} R(k - 16 q - 15) . E_q,     R(n) = first row of M^n.
// ~60 vector instructions per step instead of the 6 x 64 of the serial recurrence -- but in another order of
// operations, so not the reference's bits: the result only SPECULATES the decisions "dev < (int)avg";
// whb_verify_kernel checks them against the exact recurrence.
struct WhbScan {
	double m2[4], m4[4], m8[4];   // M^2, M^4, M^8 (m11, m12, m21, m22), wave-uniform
	double cy1, cy2;              // R(k + 1)
	double ce[3][2];              // R(k - 16 q - 15), zero where q >= k / 16
	double a1;
};
__device__ __forceinline__ void whb_scan_init(WhbScan &w, double a1, double a2, int ln)
{
	w.a1 = a1;
	w.cy1 = w.cy2 = 0.0;
#pragma unroll
	for (int q = 0; q < 3; q++)
		w.ce[q][0] = w.ce[q][1] = 0.0;
	// g(n): impulse response of 1 / (1 - a1 z^-1 - a2 z^-2); M^n = [[g(n), a2 g(n-1)], [g(n-1), a2 g(n-2)]]
	double gm2 = 0.0, gm1 = 0.0, g = 1.0;  // g(n-2), g(n-1), g(n) at n = 0 (g(-1) = 0; g(-2) only enters as a2 g(-2) = 1 at n = 1)
	for (int n = 0; n <= 64; n++) {
		if (n == 2 || n == 4 || n == 8) {
			double *m = n == 2 ? w.m2 : (n == 4 ? w.m4 : w.m8);
			m[0] = g;
			m[1] = a2 * gm1;
			m[2] = gm1;
			m[3] = a2 * gm2;
		}
		if (n == ln + 1) {
			w.cy1 = g;
			w.cy2 = a2 * gm1;
		}
#pragma unroll
		for (int q = 0; q < 3; q++)
			if (n >= 1 && n == ln - 16 * q - 15) {
				w.ce[q][0] = g;
				w.ce[q][1] = a2 * gm1;
			}
		const double gn = a1 * g + a2 * gm1;
		gm2 = gm1;
		gm1 = g;
		g = gn;
	}
}
// x: the lane's filter input b0 * (d(k) + 2 d(k-1) + d(k-2)); (y1, y2): the two outputs before the step
__device__ __forceinline__ double whb_scan_step(const WhbScan &w, double x, double y1, double y2)
{
	const double xs = row_shr_f64<1>(x);
	double u = __builtin_fma(w.a1, xs, x), v = xs;
	{
		const double us = row_shr_f64<2>(u), vs = row_shr_f64<2>(v);
		const double un = __builtin_fma(w.m2[0], us, __builtin_fma(w.m2[1], vs, u));
		v = __builtin_fma(w.m2[2], us, __builtin_fma(w.m2[3], vs, v));
		u = un;
	}
	{
		const double us = row_shr_f64<4>(u), vs = row_shr_f64<4>(v);
		const double un = __builtin_fma(w.m4[0], us, __builtin_fma(w.m4[1], vs, u));
		v = __builtin_fma(w.m4[2], us, __builtin_fma(w.m4[3], vs, v));
		u = un;
	}
	{
		const double us = row_shr_f64<8>(u), vs = row_shr_f64<8>(v);
		const double un = __builtin_fma(w.m8[0], us, __builtin_fma(w.m8[1], vs, u));
		v = __builtin_fma(w.m8[2], us, __builtin_fma(w.m8[3], vs, v));
		u = un;
	}
	double y = __builtin_fma(w.cy1, y1, __builtin_fma(w.cy2, y2, u));
#pragma unroll
	for (int q = 0; q < 3; q++) {
		const double eu = readlane_f64(u, 16 * q + 15), ev = readlane_f64(v, 16 * q + 15);
		y = __builtin_fma(w.ce[q][0], eu, __builtin_fma(w.ce[q][1], ev, y));
	}
	return y;
}

// REDO (EXACT only): launched behind whb_verify_kernel over all streams, does the submit of those it failed again.
template <bool EXACT, bool REDO>
__global__ __launch_bounds__(64) void whb_demod_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						       const int32_t *__restrict__ dev32, int n_streams, int n_blocks,
						       long long sample_base, ChainLaunch L, int a, WinTables T,
						       tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags)
{
	constexpr bool redo = REDO;
	static_assert(EXACT || !REDO, "only the exact kernel redoes a submit");
	extern __shared__ __attribute__((aligned(16))) uint8_t rdata_lds[];  // 64 x 64 B, used by the decoder tail
	// Wave priority 1: since the check stopped being the longest kernel of the batch (round 4) this one is, and its 1024
	// statically placed waves end with the slowest: 5.5 -> 5.1 ms inside the batch, the batch 1 % shorter
	// (profiles/r04_ab_whb_prio.txt; priority 2: the same).
#ifndef TFREC_AMD_WHB_PRIO
#define TFREC_AMD_WHB_PRIO 1
#endif
	__builtin_amdgcn_s_setprio(TFREC_AMD_WHB_PRIO);
	// One of these waves per SIMD, never two: the kernel claims 264 of a SIMD's 512 registers (256 + 8 accumulation
	// registers it never touches).  Its one-wave workgroups are dispatched while the other chains' kernels fill the chip
	// and land wherever a wave slot is free; two of them on one SIMD share its VALU (the recurrence alone wants 3/4 of
	// it) and run at half speed, and the kernel ends with its slowest stream: a third of the streams ran doubled up,
	// the slowest took 2.2x the average (profiles/ubench/whb_cycles.py span); with the claim 8.7 -> 6.7 ms in the batch.
#ifndef TFREC_AMD_WHB_THIN
	if (EXACT && !REDO)  // (the redo launch: a thousand workgroups that return at once must not wait for half a SIMD each)
		asm volatile("" ::: "v255", "a7");
#endif
	constexpr int kStep = 64;  // samples per iteration: one per lane
	const int ln = threadIdx.x;
	// one wave per stream (the body returns where the stream has nothing more to do)
	auto stream_body = [&](const int s) {
	uint8_t *const rdata_wave = rdata_lds;
	const int c = a * n_streams + s;
	const int M = n_blocks * kBlockDec;
	const int count = T.count[c];
	constexpr int kStateChunks = (int)(sizeof(ChainState) / 16);
	static_assert(kStateChunks <= 64, "a wave copies a ChainState in one go");
	if (!EXACT) {
		// what a redo of this submit would start from (whb_verify_kernel decides): the generation first, then the state
		const uint32_t gen = __atomic_load_n(&T.whbgen[s], __ATOMIC_RELAXED);
		__threadfence();
		if (ln < kStateChunks)
			reinterpret_cast<uint4 *>(&T.whbsnap[s])[ln] = reinterpret_cast<const uint4 *>(&L.states[a][s])[ln];
		if (ln == 0)
			T.whbseen[s] = gen;
	} else if (redo) {
		// ---- the stream's speculative pass over this submit did not reproduce the exact recurrence (or started from a state
		// a redo has replaced since): retract its events, restore the state it should have started from, and run the
		// submit again with the exact recurrence
		if (!T.whbfail[s])
			return;
		for (int j = ln; j < count; j += 64) {
			const int idx = T.result[(size_t)c * T.cap + j].first_cand_g;
			if (T.result[(size_t)c * T.cap + j].closed && idx >= 0 && (uint32_t)idx < eb->capacity) {
				events[idx].status = (uint8_t)kStatusDead;
				atomicAdd(&eb->dead, 1u);
			}
		}
		// The redo launch's L.states[a] is the context's PRIVATE scratch array (T.whbscr): the speculative kernels of the
		// submits behind this one read and write the live state (T.whbpub) in place while this runs for milliseconds.
		const bool stale = T.whbseen[s] != T.whbgen[s];
		const ChainState *from = stale ? &T.whbX[s] : &T.whbsnap[s];
		if (ln < kStateChunks)
			reinterpret_cast<uint4 *>(&L.states[a][s])[ln] = reinterpret_cast<const uint4 *>(from)[ln];
		__threadfence();
		__syncthreads();
		if (!stale && ln == 0) {  // the filter's exact state at the submit's start (the snapshot holds the speculated one)
			const WhbExact x = T.whbx0[s];
			ChainState &st0 = L.states[a][s];
			st0.iir_avg.yn = x.y1;
			st0.iir_avg.yn1 = x.y2;
			st0.iir_avg.dn1 = 0.5 * (double)x.fd1;
			st0.iir_avg.dn2 = 0.5 * (double)x.fd2;
			// a locked window open at the submit's start: the snapshot froze the SPECULATED integer, the check accepted it as
			// the exact one's neighbour (carry = exact - speculated, 0 unless such a window is open) -- the exact kernel must
			// continue the window with the exact integer (whb.cpp:653-654)
			st0.avg_of += x.carry;
		}
		__threadfence();
		__syncthreads();
	}
#ifdef TFREC_AMD_PROFILE_WHB
	long long pf_rec = 0, pf_steps = 0, pf_usteps = 0, pf_t0 = __builtin_readcyclecounter();
	long long pf_top = 0, pf_walk = 0, pf_tail = 0, pf_mark = 0;
	const long long pf_w0 = wall_clock64();  // 100 MHz
#endif
	if (count > 0) {
		const uint32_t *drow = dec + (size_t)s * dec_stride;
		const int32_t *dvrow = dev32 + (size_t)s * T.slots * 32;
		const ChainParams &p = L.params[a];
		ChainState &st = L.states[a][s];
		const double a1 = p.iir_avg.a1, a2 = p.iir_avg.a2;
		const double bh = 0.5 * p.iir_avg.b0;  // t = fl(b0 * (0.5 * dev)) = fl((b0 / 2) * dev): scaling by two is exact
		// Samples per bit: the reference builds its one whb_demod with (1536000 / 4.0) / 6000 = 64.0 (main.cpp:217) and the
		// C ABI has no other (capi.hip: reg[]; tfrec_amd_create rejects a WHB chain whose spb differs from kWhbSpb).  As a
		// constant, (int)((tdiff + spb / 2) / spb) (whb.cpp:668) is a shift, "tdiff > 3 * spb / 4" (:664) is "tdiff >= 49",
		// and a 64-sample step holds at most TWO accepted candidates, the second of which (tdiff in [49, 63]) emits one bit.
		constexpr int tmin = 3 * kWhbSpb / 4 + 1;  // smallest integer tdiff with tdiff > 3*spb/4 (whb.cpp:664)
		static_assert(kWhbSpb == 64 && (1 << kWhbSpbShift) == kWhbSpb && tmin > kStep / 2 && (kStep - 1 + kWhbSpb / 2) >> kWhbSpbShift == 1,
			      "the candidate walk knows two candidates per step, the second one bit long");
		// ---- per-stream state, wave-uniform.  The state arrives through vector loads; v_readfirstlane moves what the
		// candidate walk computes with into scalar registers (round 6: the compiler kept `synced`, the byte counters and
		// the descrambler history in vector registers and paid a vector compare + branch on vcc for every test of them)
		auto sgpr = [](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
		auto sgpr64 = [](long long v) -> long long {
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(unsigned long long)v);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)v >> 32));
			return (long long)(((unsigned long long)hi << 32) | lo);
		};
		double y1 = st.iir_avg.yn, y2 = st.iir_avg.yn1;  // iir_avg: its last two outputs ...
		// ... and its last two inputs: 0.5 * (a stage-1 output) each, carried as the integers
		int fd1 = sgpr((int)(2.0 * st.iir_avg.dn1)), fd2 = sgpr((int)(2.0 * st.iir_avg.dn2));
		int avg_of = sgpr(st.avg_of), last_dev = sgpr(st.last_dev);
		long long step0 = sgpr64((long long)st.step);               // samples since the window opened, at the window's first sample here
		// ... since the last accepted candidate, at the step's first sample: `int tdiff = step - last_peak` (whb.cpp:659) keeps
		// the low 32 bits of the difference, and so does this (unsigned: the additions may wrap)
		uint32_t since = (uint32_t)sgpr((int)(uint32_t)(st.step - st.last_peak));
		// whb.cpp:678 sums I*I+Q*Q of the synced samples in a double.  The sums are integers far below 2^53, so the
		// additions are exact in any order: the wave sums a window's samples once, at its end (power_sum).
		double rssi_d = st.rssi_d;    // rssi collected in earlier submits of a still-open window
		int synced = sgpr(st.synced);
		// the decoder registers the sync search depends on (store_bit always leaves last_psk == psk, so nrzs toggles
		// exactly when the bit differs from the previous one: nrzs(t) = bit(t) ^ K with K fixed for the stream)
		uint32_t srr = (uint32_t)sgpr((int)__brev(st.sr));  // whb_decoder::sr, newest bit at the LSB
		const uint32_t kmask = (uint32_t)sgpr((st.nrzs ^ st.w_last_bit) & 1 ? -1 : 0);
		// history of the emitted BITS, newest at the LSB: whb_decoder::lfsr (the history of nrzs, whb.cpp:579) is bhist ^ kmask
		uint32_t bhist = (uint32_t)sgpr((int)st.lfsr) ^ kmask;
		// sr_cnt / byte_cnt while the decoder has not locked since its last flush (they only matter before a stream's
		// first flush, when the zero-initialised sr_cnt = 0 lets store_bit count bytes without a sync word)
		int sc = sgpr(st.sr_cnt), bc = sgpr(st.byte_cnt);
		const bool cont = T.cont[c] != 0;
		// EXACT = false: the filter's steps are evaluated lane-parallel (whb_scan_step) and their decisions recorded for
		// whb_verify_kernel: one word per step in which the filter ran, numbered through the submit
		WhbScan scan;
		if (!EXACT)
			whb_scan_init(scan, a1, a2, ln);
		// A candidate test against the frozen average is AMBIGUOUS if it would come out differently with the average up to
		// `tol` higher or lower: avg_of - dev in [-tol + 1, tol].  tol = 1 (the speculated (int) may be the exact one's
		// neighbour); tests widen it and perturb the frozen integer (WinTables::whb_test_perturb).
		const int perturb = EXACT ? 0 : whb_hook_perturb(T);
		const int amb_tol = perturb > 1 ? perturb : (perturb < -1 ? -perturb : 1);
		const int amb_lo = amb_tol - 1;
		const uint32_t amb_w = 2u * (uint32_t)amb_tol;
		WhbStepRec *const recrow = T.whbrec + (size_t)s * T.whbrec_stride;
		int vstep = 0;
		// ... and the filter's input sequence (whb_check.h: the exact chain walks it a stream per lane): the stage-1 outputs of
		// the samples the average ran on, in order, behind each other
		int32_t *const dense = T.whbdense + (size_t)s * T.whbdense_stride;
		int dcount = 0;

		// feed `len` emitted bits (bit i of `e` = i-th bit, len <= 32) to the sync search; true if sr hit the sync word.
		// The descrambled bit is nrzs(t) ^ nrzs(t-12) ^ nrzs(t-17) (whb.cpp:578) = b(t) ^ b(t-12) ^ b(t-17) ^ K.
		auto feed = [&](uint32_t e, int len) -> bool {
			const uint32_t emask = len >= 32 ? ~0u : (1u << len) - 1u;
			const uint32_t brun = __brev(e & emask) >> (32 - len);                   // the run's bits, newest at the LSB
			const unsigned long long hb = ((unsigned long long)bhist << len) | brun;
			const uint32_t orun = ((uint32_t)(hb ^ (hb >> 12) ^ (hb >> 17)) ^ kmask) & emask;  // descrambled bits
			const unsigned long long sv = ((unsigned long long)srr << len) | orun;
			const bool hit = ln < len && (uint32_t)(sv >> (len - 1 - (ln < len ? ln : 0))) == kWhbSyncRev;
			bhist = (uint32_t)hb;
			srr = (uint32_t)sv;
			return __ballot(hit) != 0ull;
		};

		for (int j = 0; j < count; j++) {
			// ---- the window
			const int og = sgpr(T.open[(size_t)c * T.cap + j]);
			const int close = sgpr(T.close[(size_t)c * T.cap + j]);
			const bool closed = close < M;
			const int n = (closed ? close : M - 1) - og + 1;
			const int nch = (n + kStep - 1) / kStep;
			const int slot0 = win_slot0(og, j);
			// the stage-1 outputs of the window's step 0 (a wave-uniform pointer: the loads take it as their scalar base and the
			// lane as their offset)
			const int32_t *wq = dvrow + (size_t)slot0 * 32;
			uint16_t *ent = reinterpret_cast<uint16_t *>(T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j);
			// kWhbAhead steps stay in flight (past the window's end: the row's next slots or the slack behind it, never used)
			int cur = wq[ln], nxt[kWhbAhead];
#pragma unroll
			for (int k = 0; k < kWhbAhead; k++)
				nxt[k] = wq[kStep * (k + 1) + ln];
			wq += kStep * (kWhbAhead + 1);  // the step the loop loads next
			if (!(j == 0 && cont)) {  // window opens: whb_demod::reset, whb.cpp:616-623
				rssi_d = 0;
				step0 = 0;
				since = 0;
			}
			const int vbase = vstep;
			int lock_pos = -1, avg_frozen = 0;  // (window-relative sample at which the decoder locked in this window)
			// a candidate test against the FROZEN average that would come out differently with the average one higher or
			// lower: only then does it matter that (int) of the speculated average may be the exact one's neighbour
			bool amb = false;
			if (ln == 0) {
				WhbStart ws;
				ws.sr = __brev(srr);
				ws.lfsr = bhist ^ kmask;
				ws.sr_cnt = sc;
				ws.byte_cnt = bc;
				ws.synced = synced;
				ws.pad_[0] = ws.pad_[1] = ws.pad_[2] = 0;
				T.whbstart[(size_t)s * T.cap + j] = ws;
			}
			// run lengths of the accepted candidates, one uint16 entry each, stored as they are produced: every lane stores the
			// same value to the same address (rounds 3-5 collected 64 of them in a lane register first: a vector compare, a
			// select and a test of the counter per entry)
			int nent = 0;
			auto put_ent = [&](uint32_t v16) {
				ent[nent] = (uint16_t)v16;
				nent++;
			};
			// whb.cpp:677-678: the power of the samples from the one the decoder locked on (or the window's first here, if it began
			// locked) to the window's last, summed at the window's end (round 6: per step it was two more loads in flight beside
			// the stage-1 outputs' and their rotation; -1 % of the batch, profiles/r06_ab_power_sum.txt)
			int rssi_from = synced ? 0 : -1;
			auto power_sum = [&](int from) -> unsigned long long {
				unsigned long long acc = 0ull;
#pragma unroll 4
				for (int m = from + ln; m < n; m += kStep) {
					const uint32_t w = drow[og + m];
					const int I = (int)(int16_t)(w & 0xffff), Q = (int)w >> 16;
					acc += (unsigned long long)(uint32_t)(I * I + Q * Q);
				}
#pragma unroll
				for (int o = 32; o >= 1; o >>= 1)
					acc += __shfl_xor(acc, o, 64);
				return acc;
			};
			for (int i = 0; i < nch; i++) {
				// ---- (1) this step's inputs; the next two steps' are in flight
				const int nv = n - kStep * i < kStep ? n - kStep * i : kStep;
				const unsigned long long valid = nv < kStep ? (1ull << nv) - 1ull : ~0ull;  // the step's samples inside the window
				const int nxn = wq[ln];
				wq += kStep;
				const int dev = cur;
				const int sh1 = wave_shr1(dev);
				const int devm1 = ln == 0 ? last_dev : sh1;  // dev > last_dev (whb.cpp:663): the sample before the step
				const unsigned long long rise_m = __builtin_amdgcn_ballot_w64(dev > devm1) & valid;
				const bool was_synced = synced != 0;
				unsigned long long mask;
				const double y1_in = y1;
				double ym = 0.0;  // the average after the lane's sample (while the decoder is unsynced)
#ifdef TFREC_AMD_PROFILE_WHB
				pf_steps++;
				pf_mark = __builtin_readcyclecounter();
#endif
				// the filter's input history and state at the step's first sample (a lock inside the step reads them again)
				const int fd1_in = fd1, fd2_in = fd2;
				const double y2_in = y2;
				// (2) + (3'): the lane-parallel evaluation of the average over the step's 64 samples from that state -- the
				// feed-forward half of iir2::step for the lane's sample (see iir_step_t: x = b0 d(k) + b1 d(k-1) + b2 d(k-2) with
				// b1 = 2 b0, b2 = b0), then whb_scan_step
				auto scan_step = [&]() -> double {
					const int devm1f = ln == 0 ? fd1_in : sh1;  // the filter's own input history (it pauses while synced)
					const int sh2 = wave_shr1(devm1f);
					const int devm2f = ln == 0 ? fd2_in : sh2;
					const double t0 = bh * (double)dev, t1 = bh * (double)devm1f, t2 = bh * (double)devm2f;
					return whb_scan_step(scan, __builtin_fma(2.0, t1, t0) + t2, y1_in, y2_in);
				};
				if (!was_synced) {
#ifdef TFREC_AMD_PROFILE_WHB
					const long long pf_a = __builtin_readcyclecounter();
					pf_usteps++;
					pf_top += pf_a - pf_mark;
#endif
					if (EXACT) {
						// ---- (2) feed-forward half of iir2::step for the lane's sample (see iir_step_t)
						const int devm1f = ln == 0 ? fd1 : sh1;  // the filter's own input history (it pauses while synced)
						const int sh2 = wave_shr1(devm1f);
						const int devm2f = ln == 0 ? fd2 : sh2;
						const double t0 = bh * (double)dev, t1 = bh * (double)devm1f, t2 = bh * (double)devm2f;
						const double ffp = __builtin_fma(2.0, t1, t0);  // P; B2 = t2
						// ---- (3) the serial feedback recurrence, 64 samples (a window's last, partial step runs it over whatever
						// follows the window: finite numbers, never looked at): whb_chain_asm.h.  The feed-forward pairs as four
						// row-replicated sets: lane 16r + i holds sample 16j + i of set j.
						// (two v_permlane16/32_swap levels per dword: no LDS round trip in the step -- the CU's LDS pipe belongs to
						// the front end's workgroups, and a lone wave waiting behind them was the slowest stream of the batch)
						double inp[4], inb[4];
						rows_replicate(ffp, inp);
						rows_replicate(t2, inb);
						double z0, z1, z2, z3, tt, tq, ya = 0.0, yb = 0.0, yc = y2, yd = y1;
						asm volatile(TFREC_WHB_CHAIN_ASM
							     : [Y0] "+v"(ya), [Y1] "+v"(yb), [Y2] "+v"(yc), [Y3] "+v"(yd), [Z0] "=&v"(z0), [Z1] "=&v"(z1),
							       [Z2] "=&v"(z2), [Z3] "=&v"(z3), [T] "=&v"(tt), [Q] "=&v"(tq)
							     : [a1] "s"(a1), [a2] "s"(a2), [ONE] "v"(1.0), [P0] "v"(inp[0]), [B0] "v"(inb[0]), [P1] "v"(inp[1]),
							       [B1] "v"(inb[1]), [P2] "v"(inp[2]), [B2] "v"(inb[2]), [P3] "v"(inp[3]), [B3] "v"(inb[3]));
						const int zq = ln & 3;
						ym = zq == 0 ? z0 : (zq == 1 ? z1 : (zq == 2 ? z2 : z3));  // y(ln)
						if (nv == kStep) {
							y1 = yd;
							y2 = yc;
						} else {  // the filter stops with the window's last sample
							y1 = readlane_f64(ym, nv - 1);
							y2 = nv > 1 ? readlane_f64(ym, nv > 1 ? nv - 2 : 0) : y1_in;
						}
#ifdef TFREC_AMD_PROFILE_WHB
							pf_mark = __builtin_readcyclecounter();
							pf_rec += pf_mark - pf_a;
#endif
					} else {
						// ---- (3') all 64 samples at once
						ym = scan_step();
						y1 = readlane_f64(ym, nv - 1);
						y2 = nv > 1 ? readlane_f64(ym, nv > 1 ? nv - 2 : 0) : y1_in;
#ifdef TFREC_AMD_PROFILE_WHB
						pf_mark = __builtin_readcyclecounter();
						pf_rec += pf_mark - pf_a;
#endif
					}
					// |0.5*dev| <= 6.6e8 and the decision-level low-pass has an L1 gain of 1.09: (int) never saturates
					const unsigned long long below = __builtin_amdgcn_ballot_w64(dev < (int)ym) & valid;
					if (!EXACT) {
						if (ln == 0) {  // (where the decoder locks in this step, the window's end rewrites meta and avgf)
							WhbStepRec r;
							r.below = below;
							r.meta = (uint32_t)(slot0 + 2 * i) | ((uint32_t)(nv - 1) << kWhbRecNvShift);
							r.avgf = 0;
							recrow[vstep] = r;
						}
						vstep++;
					}
					mask = below & rise_m;
				} else {
					mask = __builtin_amdgcn_ballot_w64(dev < avg_of) & rise_m;
					if (!EXACT)
						amb = amb || (__builtin_amdgcn_ballot_w64((uint32_t)(avg_of + amb_lo - dev) < amb_w) & rise_m) != 0ull;
				}
				// ---- (4) accepted candidates
				int locked_at = -1;
				// one accepted candidate at sample k of the step, tdiff samples after the one before it (whb.cpp:665-674);
				// ONE = std::true_type: the step's second candidate, whose run is one bit long
				auto pulse = [&](const int k, const int tdiff, auto ONE) {
					constexpr bool one = decltype(ONE)::value;
					// whb.cpp:666-673: one 0, then (bit0 - 1) ones
					const int bit0 = one ? 1 : (tdiff + kWhbSpb / 2) >> kWhbSpbShift;
					const int len = bit0 > 1 ? bit0 : 1;
					if (one || len < kWhbRunEsc) {
						put_ent((uint32_t)len);
					} else {
						put_ent((uint32_t)kWhbRunEsc);
						put_ent((uint32_t)len & 0xffffu);
						put_ent((uint32_t)len >> 16);
					}
					// The run "0,1,1,.." joins the bit history, in scalar registers (its first 32 bits; the ones beyond are the rare
					// tail below).  The sync search -- only while the decoder is unsynced: once it has locked, a second hit of the
					// sync word matters to the decoder stage alone, which replays the runs bit by bit -- looks at the run's
					// positions one per lane.
					const int l0 = one ? 1 : (len < 32 ? len : 32);
					const unsigned long long hb = ((unsigned long long)bhist << l0) | ((1ull << (l0 - 1)) - 1ull);
					bool hit = false;
					if (synced == 0) {
						const uint32_t emask = (uint32_t)((1ull << l0) - 1ull);
						const uint32_t orun = ((uint32_t)hb ^ (uint32_t)(hb >> 12) ^ (uint32_t)(hb >> 17) ^ kmask) & emask;
						const unsigned long long sv = ((unsigned long long)srr << l0) | orun;
						if (one) {
							hit = (uint32_t)sv == kWhbSyncRev;
						} else {  // lane l < l0: sr after all but the run's last l bits
							const unsigned long long hits = __builtin_amdgcn_ballot_w64((uint32_t)(sv >> ln) == kWhbSyncRev);
							hit = (hits & (unsigned long long)emask) != 0ull;
						}
						srr = (uint32_t)sv;
					}
					bhist = (uint32_t)hb;
					if (!one && len > 32)  // (cut to the run's first 160 bits: the registers reach a fixed point after 17 + 32 equal bits)
						for (int rest = (len > 160 ? 160 : len) - 32; rest > 0; rest -= 32)
							hit = feed(~0u, rest < 32 ? rest : 32) || hit;
					if (synced == 0) {
						if (sc >= 0) {  // sr_cnt / byte_cnt over `len` bits without a sync word (whb.cpp:590-596)
							const int i0 = (8 - sc) & 7;  // first bit of the run that finds sr_cnt == 0
							bc += i0 < len ? (len - 1 - i0) / 8 + 1 : 0;
							sc = (sc + len) & 7;
						}
						if (hit) {  // the decoder locked at sample k: the average stops after it (whb.cpp:653)
							synced = 1;
							locked_at = k;
							const double yk = readlane_f64(ym, k), ykm1 = readlane_f64(ym, k > 0 ? k - 1 : 0);
							const int dk = __builtin_amdgcn_readlane(dev, k);
							const int dkm1 = __builtin_amdgcn_readlane(dev, k > 0 ? k - 1 : 0);
							y2 = k > 0 ? ykm1 : y1_in;
							y1 = yk;
							fd2 = k > 0 ? dkm1 : fd1;
							fd1 = dk;
							avg_of = (int)yk + perturb;
							lock_pos = kStep * i + k;
							avg_frozen = avg_of;
							// the rest of the step's candidates against the frozen avg_of
							const unsigned long long after = k < kStep - 1 ? ~0ull << (k + 1) : 0ull;
							mask = __builtin_amdgcn_ballot_w64(dev < avg_of) & rise_m & after;
							if (!EXACT)
								amb = amb || (__builtin_amdgcn_ballot_w64((uint32_t)(avg_of + amb_lo - dev) < amb_w) & rise_m & after) != 0ull;
						}
					}
				};
				if (mask) {
					// first k with tdiff = since + k > 3*spb/4 (whb.cpp:664), in the reference's int arithmetic
					// (a difference that has wrapped to a negative int accepts nothing, as in the reference)
					const int kmin = (int)since < -kStep ? kStep : tmin - (int)since;
					const unsigned long long m1 = kmin > 0 ? (kmin > kStep - 1 ? 0ull : mask & (~0ull << kmin)) : mask;
					if (m1) {
						const int k = __builtin_ctzll(m1);
						pulse(k, (int)(since + (uint32_t)k), std::false_type{});
						since = (uint32_t)-k;  // last_peak = this sample
						const int k2min = k + tmin;
						// (`mask` again: a lock at k replaced it by the tests against the frozen average)
						const unsigned long long m2 = k2min > kStep - 1 ? 0ull : mask & (~0ull << k2min);
						if (m2) {
							const int k2 = __builtin_ctzll(m2);
							pulse(k2, k2 - k, std::true_type{});
							since = (uint32_t)-k2;
						}
					}
				}
#ifdef TFREC_AMD_PROFILE_WHB
				{
					const long long t = __builtin_readcyclecounter();
					pf_walk += t - pf_mark;
					pf_mark = t;
				}
#endif
				// ---- (5) the step's state
				const int dl1 = __builtin_amdgcn_readlane(dev, nv - 1);
				if (!EXACT && !was_synced) {  // the samples of this step the average ran on: up to the lock, or all of them
					const int nvf = locked_at >= 0 ? locked_at + 1 : nv;
					if (ln < nvf)
						dense[dcount + ln] = dev;
					dcount += nvf;
				}
				if (!was_synced && locked_at < 0) {  // the whole step went through the average
					fd2 = nv > 1 ? __builtin_amdgcn_readlane(dev, nv > 1 ? nv - 2 : 0) : fd1;
					fd1 = dl1;
					avg_of = (int)y1;
				}
				last_dev = dl1;
				since += (uint32_t)nv;
				if (locked_at >= 0)
					rssi_from = kStep * i + locked_at;
				cur = nxt[0];
#pragma unroll
				for (int k = 0; k + 1 < kWhbAhead; k++)
					nxt[k] = nxt[k + 1];
				nxt[kWhbAhead - 1] = nxn;
#ifdef TFREC_AMD_PROFILE_WHB
				pf_tail += __builtin_readcyclecounter() - pf_mark;
#endif
			}
			// ---- the window's last sample in this submit
			WinResult res;
			res.nbits = nent;
			res.closed = 0;
			long long rssi_out = 0;
			if (closed) {  // timeout_cnt reached 0, whb.cpp:691-702
				if (synced) {
					const unsigned long long tot = power_sum(rssi_from);
					(void)feed(0u, 16);  // 16 x store_bit(0); the flush then clears sr and synced (whb.cpp:559-563)
					rssi_out = (long long)(rssi_d + (double)tot);
					res.closed = 1;
					srr = 0;
					synced = 0;
					sc = -1;
					bc = 0;
				}
				rssi_d = 0;
				step0 = 0;
				since = 0;
			} else {  // the window continues in the next submit
				if (synced)
					rssi_d += (double)power_sum(rssi_from);
				step0 += n;
			}
			res.rssi_i = (int32_t)(uint32_t)((unsigned long long)rssi_out & 0xffffffffull);
			res.offset = (int32_t)(uint32_t)((unsigned long long)rssi_out >> 32);
			res.lbi_out = 0;
			res.first_cand_g = -1;
			// for whb_verify_kernel: the filter steps of this window (their records start at mark_lvl), where the decoder
			// locked (window-relative sample, -1: it did not), the average it froze there, and (last_bit) whether a candidate
			// test of this window would change with that average off by one
			res.bitcnt = vstep - vbase;
			res.dmax = lock_pos;
			res.dmin = avg_frozen;
			res.mark_lvl = vbase;
			res.last_bit = amb ? 1 : 0;
			res.resume = -1;
			if (ln == 0)
				T.result[(size_t)c * T.cap + j] = res;
			if (!EXACT) {
				// whb_verify_kernel's view of the window's end: the filter's run ended with a lock (the step's record says on
				// which sample, what was frozen, and whether the rest of the window could tell it from its neighbours), or the
				// window never ran the filter (it began locked: one record without a step)
				const uint32_t wfl = (amb ? kWhbRecAmb : 0u) | (res.closed ? kWhbRecClosed : 0u);
				if (lock_pos >= 0) {
					if (ln == 0) {
						WhbStepRec *r = &recrow[vbase + (lock_pos >> 6)];
						r->meta = (uint32_t)(slot0 + 2 * (lock_pos >> 6)) | ((uint32_t)(lock_pos & 63) << kWhbRecNvShift) | kWhbRecLock | wfl;
						r->avgf = avg_frozen;
					}
				} else if (vstep == vbase) {
					if (ln == 0) {
						WhbStepRec r;
						r.below = 0ull;
						r.meta = kWhbRecPseudo | wfl;
						r.avgf = 0;
						recrow[vstep] = r;
					}
					vstep++;
				}
			}
		}
		if (!EXACT && ln == 0) {
			WhbStepRec r;
			r.below = 0ull;
			r.meta = kWhbRecEnd;
			r.avgf = 0;
			recrow[vstep] = r;
			T.whbdense_n[s] = dcount;
		}
		if (ln == 0) {
			const uint32_t lw = drow[M - 1];
			st.prev_i = (int)(int16_t)(lw & 0xffff);
			st.prev_q = (int)lw >> 16;
			st.timeout_cnt = T.timeout_next[c];
			st.last_dev = last_dev;
			st.avg_of = avg_of;
			st.step = (unsigned long long)step0;
			st.last_peak = (unsigned long long)(step0 - since);
			st.rssi_d = rssi_d;
			st.iir_avg.yn = y1;
			st.iir_avg.yn1 = y2;
			st.iir_avg.dn1 = 0.5 * (double)fd1;
			st.iir_avg.dn2 = 0.5 * (double)fd2;
		}
	} else if (ln == 0) {  // no window in this submit: only the carried sample and timeout move on
		if (!EXACT) {
			WhbStepRec r;
			r.below = 0ull;
			r.meta = kWhbRecEnd;
			r.avgf = 0;
			T.whbrec[(size_t)s * T.whbrec_stride] = r;
			T.whbdense_n[s] = 0;
		}
		ChainState &st = L.states[a][s];
		const uint32_t lw = dec[(size_t)s * dec_stride + M - 1];
		st.prev_i = (int)(int16_t)(lw & 0xffff);
		st.prev_q = (int)lw >> 16;
		st.timeout_cnt = T.timeout_next[c];
	}
#ifdef TFREC_AMD_PROFILE_WHB
	if (ln == 0) {  // cycles: recurrence | whole demodulator; steps: all | with the recurrence
#ifndef TFREC_AMD_PROFILE_WHB_SPAN
		atomicAdd(&T.stats[5], (unsigned long long)pf_rec);
#endif
#ifdef TFREC_AMD_PROFILE_WHB_SPAN  // of the sixth submit: earliest / latest workgroup start, latest end (100 MHz ticks), sum of starts
		if (sample_base == 5LL * n_blocks * kBlockDec) {
			atomicMax(&T.stats[1], ~(unsigned long long)pf_w0);
			atomicMax(&T.stats[2], (unsigned long long)pf_w0);
			atomicMax(&T.stats[3], (unsigned long long)wall_clock64());
			atomicAdd(&T.stats[0], (unsigned long long)pf_w0 & 0xffffffffffull);
			// the slowest stream: its cycles (high 40 bits) and steps (low 24)
			atomicMax(&T.stats[5], ((unsigned long long)(__builtin_readcyclecounter() - pf_t0) << 24) | (unsigned long long)pf_steps);
			{  // histogram of the streams' cycles per step (x100), 5 buckets of 12 bits: < 25, < 30, < 35, < 45, more
				const long long cps = (__builtin_readcyclecounter() - pf_t0) / (pf_steps > 0 ? pf_steps : 1) / 100;
				const int b = cps < 25 ? 0 : (cps < 30 ? 1 : (cps < 35 ? 2 : (cps < 45 ? 3 : 4)));
				atomicAdd(&T.stats[6], 1ull << (12 * b));
			}
		}
#else
		atomicAdd(&T.stats[1], (unsigned long long)pf_top);
		atomicAdd(&T.stats[2], (unsigned long long)pf_walk);
		atomicAdd(&T.stats[3], (unsigned long long)pf_tail);
#endif
		atomicAdd(&T.stats[7], (unsigned long long)(__builtin_readcyclecounter() - pf_t0));
#ifndef TFREC_AMD_PROFILE_WHB_SPAN
		atomicAdd(&T.stats[6], (unsigned long long)(wall_clock64() - pf_w0));
#endif
		atomicAdd(&T.stats[4], (unsigned long long)((pf_steps << 32) | pf_usteps));
	}
#endif
	// ---- decoder tail: the stream's windows, one per lane, then the stream's commit (lane 0)
	__threadfence();  // the runs, results and start registers
	__syncthreads();
	for (int j = ln; j < count; j += 64)
		whb_decode_window(s, j, n_streams, L, a, T, rdata_wave + 64 * ln);
	__threadfence();
	__syncthreads();
	if (ln == 0)
		whb_commit_stream(s, n_streams, n_blocks, sample_base, L, a, T, events, eb, flags, rdata_wave);
	if (EXACT && redo) {
		// Publish the private copy: to whbX (what later redos of stale submits start from) and to the live state -- a
		// speculative kernel that STARTS after the generation counter moved reads it and is not stale; one that started
		// before (or is writing the live state right now) saw the old generation and will be redone from whbX whatever it
		// reads or leaves behind.  State first, then the fence, then the counter.
		__threadfence();
		__syncthreads();
		// ... except ChainState::iir: the stage-1 low-pass state belongs to the biquad stage (fix_chain), which has carried it
		// on through the submits behind this one while the redo ran -- a whole-state copy (as the in-place restore of round 3
		// was) puts a value of several submits ago back and every later stage-1 output of the stream is wrong
		constexpr int kIirChunk0 = (int)(offsetof(ChainState, iir) / 16), kIirChunk1 = (int)(offsetof(ChainState, iir_avg) / 16);
		static_assert(offsetof(ChainState, iir) % 16 == 0 && offsetof(ChainState, iir_avg) % 16 == 0, "ChainState::iir must fill whole 16-byte chunks");
		if (ln < kStateChunks) {
			const uint4 v = reinterpret_cast<const uint4 *>(&L.states[a][s])[ln];
			reinterpret_cast<uint4 *>(&T.whbX[s])[ln] = v;
			if (ln < kIirChunk0 || ln >= kIirChunk1)
				reinterpret_cast<uint4 *>(&T.whbpub[s])[ln] = v;
		}
		__threadfence();
		__syncthreads();
		if (ln == 0) {
			const ChainState &st1 = L.states[a][s];
			WhbExact x;
			x.y1 = st1.iir_avg.yn;
			x.y2 = st1.iir_avg.yn1;
			x.fd1 = (int)(2.0 * st1.iir_avg.dn1);
			x.fd2 = (int)(2.0 * st1.iir_avg.dn2);
			x.carry = x.pad_ = 0;
			T.whbx[s] = x;
			T.whbfail[s] = 0;
			__threadfence();
			atomicAdd(&T.whbgen[s], 1u);
			atomicAdd(&T.stats[6], 1ull);
		}
	}
	};
	if (!REDO) {
		stream_body((int)blockIdx.x);
	} else {
		// The redo launch: a handful of workgroups look through the streams' flags, 64 at a time, and redo the failed ones
		// one after the other (normally none).  As a workgroup per stream it was 1024 waves of 256 registers that had to
		// find half a SIMD each just to return: 1.2 ms per batch on the stream that sets the period.
		for (int base = 64 * (int)blockIdx.x; base < n_streams; base += 64 * (int)gridDim.x) {
			unsigned long long m = __ballot(base + ln < n_streams && T.whbfail[base + ln] != 0);
			while (m) {
				const int k = __builtin_ctzll(m);
				m &= m - 1;
				stream_body(base + k);
				__syncthreads();
			}
		}
	}
}


// ------------------------------------------------------------------------------------------------ K4v WHB verify
// whb_demod_kernel<false> takes its decisions "dev < (int)avg" (whb.cpp:662) from a lane-parallel evaluation of the
// decision-level average -- the same filter in another order of operations, ~5e-3 away from the reference's doubles
// (both accumulate their own rounding errors over the filter's 7500-sample memory), which can only matter where the
// average lies that close to dev + 1.  Here the reference's own recurrence (iir2::step in its normative association,
// the hand-scheduled chain of whb_chain_asm.h) runs over exactly the samples the demodulator ran the filter on, and
// every recorded decision is compared with it: FOUR STREAMS PER WAVE, one per row of 16 lanes.  The chain is serial
// per stream and costs a wave ~35 cycles per sample whatever its lanes hold (6 fp64 instructions, DPP-broadcast inputs):
// executed for ONE stream per wave, as the exact demodulator kernel does, it is a third of the batch's vector
// instructions; a row of 16 lanes is all the broadcast needs.  (A lane per stream was tried first: its arithmetic is 45
// cycles per sample, profiles/ubench/verify_chain.hip, but the flat loop around it -- 16-byte accesses of 64 different rows
// per instruction, lanes in different groups of a half-step -- ran at 108; profiles/NOTES.md round 3.)
// The code below is written per lane; the lanes of a row hold the same stream, window and step throughout, so every
// branch is uniform per row.  Where the decoder locked, the average was frozen as an integer (whb.cpp:653-654): (int) of
// the speculated double is the exact one's neighbour once in ~200 locks; that is accepted iff no candidate test of the
// window could tell the two apart (WinResult::last_bit, tracked by the demodulator kernel).
// All equal (the rule): what whb_demod_kernel<false> emitted is the reference's result, and the exact filter state is
// carried on in T.whbx.  Otherwise T.whbfail[s] is set: the stream's submit is redone by the exact kernel.
template <int N>
__device__ __forceinline__ int row_ror_i32(int v)  // lane i of a row <- lane (i - N) & 15 of the same row
{
	return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, true);
}
// the value of lane `src` (0..15, the same for all lanes of a row) of the own row
__device__ __forceinline__ int row_pick_i32(int v, int src)
{
	return __builtin_amdgcn_ds_bpermute(4 * (((int)threadIdx.x & 48) + src), v);
}
__device__ __forceinline__ double row_pick_f64(double v, int src)
{
	return __hiloint2double(row_pick_i32(__double2hiint(v), src), row_pick_i32(__double2loint(v), src));
}

// The walk is FLAT: whb_demod_kernel<false> leaves one WhbStepRec per filtered step, in order, with the position of the
// step's stage-1 outputs in it (plus a record per window that never ran the filter and an end mark), so a row's loads are
// independent of the window structure and are queued kVerAhead steps ahead (records twice as far): inside the batch the
// kernel used to spend a third of its time waiting for the ONE step it had in flight (6.2 ms against 4.2 ms alone).
#ifndef TFREC_AMD_VER_AHEAD
#define TFREC_AMD_VER_AHEAD 2
#endif
constexpr int kVerAhead = TFREC_AMD_VER_AHEAD;  // steps whose stage-1 outputs are in flight
constexpr int kVerRecAhead = 2 * kVerAhead;  // records in flight (a step's loads need its record)
static_assert(kVerRecAhead + 1 <= kWhbRecSlack, "the record prefetch stays inside the row's slack");

__global__ __launch_bounds__(256) TFREC_LAT_VGPR_ATTR void whb_verify_kernel(const int32_t *__restrict__ dev32, int n_streams, int n_blocks,
							ChainLaunch L, int a, WinTables T, int *__restrict__ carry_io)
{
	// Wave priority 0: with its loads queued ahead the check no longer sits out memory latency, it issues at the full rate
	// of its dependent chain (two thirds of a SIMD's vector cycles).  At priority 1 the waves of the other chains that share
	// its 256 SIMDs fell behind, and their kernels end with their slowest wave: the batch 3 % longer (profiles/r04_ab_verify.txt).
#ifdef TFREC_AMD_VERIFY_PRIO
	__builtin_amdgcn_s_setprio(TFREC_AMD_VERIFY_PRIO);
#endif
	// Workgroups of FOUR waves (independent: no barrier, no shared memory): a workgroup lands on one CU, a wave on each of
	// its SIMDs.  As 256 one-wave workgroups the check sat on ONE SIMD of every CU of the chip, and the four-wave workgroups
	// of the front end and the discriminator pass ran at the pace of their wave on that SIMD (profiles/NOTES.md round 3).
	const int ln = threadIdx.x & 63, row = ln >> 4, li = ln & 15;
	const int s = (blockIdx.x * 4 + ((int)threadIdx.x >> 6)) * 4 + row;
	const bool active = s < n_streams;
	const int sc_ = active ? s : 0;
	const ChainParams &p = L.params[a];
	const double a1 = p.iir_avg.a1, a2 = p.iir_avg.a2, bh = 0.5 * p.iir_avg.b0;
	const int32_t *dvrow = dev32 + (size_t)sc_ * T.slots * 32 + li;
	const uint4 *recrow = reinterpret_cast<const uint4 *>(T.whbrec + (size_t)sc_ * T.whbrec_stride);
	const uint32_t max_slot = (uint32_t)T.slots - 2u;  // (records past the end mark hold anything: their loads stay inside the row)
	WhbExact st = T.whbx[sc_];
	double y1 = st.y1, y2 = st.y2;
	int fd1 = st.fd1, fd2 = st.fd2;
	int carry = carry_io[sc_];  // exact minus speculated frozen average of a window still open and locked (0, +1, -1)
	const int carry_in = carry;
	const int tp_ = whb_hook_perturb(T);
	const int tol = tp_ > 1 ? tp_ : (tp_ < -1 ? -tp_ : 1);
	bool bad = false, done = !active;
	// ---- the rings: R[k] = record of step v + k, D[k][q] = the lane's samples 16 q + li of step v + k
	uint4 R[kVerRecAhead + 1];
	int D[kVerAhead + 1][4];
	auto samples_of = [&](const uint4 &r, int (&buf)[4]) {
		uint32_t slot = r.z & kWhbRecOffMask;
		slot = slot < max_slot ? slot : max_slot;
		const int32_t *src = dvrow + (size_t)slot * 32;
#pragma unroll
		for (int q = 0; q < 4; q++)
			buf[q] = src[16 * q];
	};
#pragma unroll
	for (int k = 0; k <= kVerRecAhead; k++)
		R[k] = recrow[k];
#pragma unroll
	for (int k = 0; k <= kVerAhead; k++)
		samples_of(R[k], D[k]);
	int v = 0;
	while (true) {
		if (__ballot(!done) == 0ull)
			break;
		if (!done) {
			const uint4 rec = R[0];
			const uint32_t meta = rec.z;
			// the loads of the steps ahead, before this step's arithmetic
			const uint4 rnew = recrow[v + kVerRecAhead + 1];
			int dnew[4];
			samples_of(R[kVerAhead + 1], dnew);
			if (meta == kWhbRecEnd) {
				done = true;
			} else if (meta & kWhbRecPseudo) {  // the window began locked (it continues one of the previous submit): no filter step
				bad = bad || (carry != 0 && (meta & kWhbRecAmb));
				if (meta & kWhbRecClosed)
					carry = 0;
			} else {
				const int nv = (int)((meta >> kWhbRecNvShift) & 63u) + 1;
				const int(&dA)[4] = D[0];
				// ---- feed-forward half of iir2::step for the lane's four samples (iir_step_t, dsp_dev.h): sample 16 q + li has
				// its predecessors in lanes li - 1, li - 2 of set q, or in the last lanes of set q - 1 (the filter's own input
				// history fd1, fd2 before the step's first sample)
				double P[4], B2[4];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const int r1 = row_ror_i32<1>(dA[q]), r2 = row_ror_i32<2>(dA[q]);
					const int e1 = q == 0 ? fd1 : row_ror_i32<1>(dA[q > 0 ? q - 1 : 0]);  // lane 15 of the set before, in lane 0
					const int e2 = q == 0 ? (li == 0 ? fd2 : fd1) : row_ror_i32<2>(dA[q > 0 ? q - 1 : 0]);  // its lanes 14, 15 in lanes 0, 1
					const int p1 = li == 0 ? e1 : r1;
					const int p2 = li < 2 ? e2 : r2;
					const double t0 = bh * (double)dA[q], t1 = bh * (double)p1;
					P[q] = __builtin_fma(2.0, t1, t0);
					B2[q] = bh * (double)p2;
				}
				// ---- the chain, 4 x 16 samples (whb_chain_asm.h): Y3 = y(-1), Y2 = y(-2) on entry, y(63), y(62) on exit; y of
				// the lane's sample 16 q + li is captured in Z[q][li & 3]
				double Y0 = 0.0, Y1 = 0.0, Y2 = y2, Y3 = y1, tt, tq, ym[4];
				const double y1_in = y1;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					double z0, z1, z2, z3;
					asm volatile(TFREC_WHB_CHAIN16_ASM
						     : [Y0] "+v"(Y0), [Y1] "+v"(Y1), [Y2] "+v"(Y2), [Y3] "+v"(Y3), [Z0] "=&v"(z0), [Z1] "=&v"(z1),
						       [Z2] "=&v"(z2), [Z3] "=&v"(z3), [T] "=&v"(tt), [Q] "=&v"(tq)
						     : [a1] "s"(a1), [a2] "s"(a2), [ONE] "v"(1.0), [P] "v"(P[q]), [B] "v"(B2[q]));
					const int zq = li & 3;
					ym[q] = zq == 0 ? z0 : (zq == 1 ? z1 : (zq == 2 ? z2 : z3));
				}
				// ---- whb.cpp:654 "(int)", :662 "dev < avg_of": the row's 64 decisions against the recorded ones
				unsigned long long word = 0;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const unsigned long long b = __ballot(16 * q + li < nv && dA[q] < (int)ym[q]);
					word |= ((b >> (16 * row)) & 0xffffull) << (16 * q);
				}
				const unsigned long long vm = nv >= 64 ? ~0ull : (1ull << nv) - 1ull;
				const unsigned long long below = ((unsigned long long)rec.y << 32) | rec.x;
				bad = bad || ((word ^ below) & vm) != 0ull;
				// ---- the filter's state after the step's last sample (nv - 1: a window's last step may be partial, and a lock
				// ends the filter's run at that sample)
				if (nv == 64) {
					y1 = Y3;
					y2 = Y2;
					fd1 = __builtin_amdgcn_update_dpp(0, dA[3], 0x150 + 15, 0xf, 0xf, true);  // row_newbcast:15
					fd2 = __builtin_amdgcn_update_dpp(0, dA[3], 0x150 + 14, 0xf, 0xf, true);
				} else {
					const int pe = nv - 1, pq = pe >> 4, pb = pe > 0 ? pe - 1 : 0, pbq = pb >> 4;
					const double ye = pq == 0 ? ym[0] : (pq == 1 ? ym[1] : (pq == 2 ? ym[2] : ym[3]));
					const double yb = pbq == 0 ? ym[0] : (pbq == 1 ? ym[1] : (pbq == 2 ? ym[2] : ym[3]));
					const int de = pq == 0 ? dA[0] : (pq == 1 ? dA[1] : (pq == 2 ? dA[2] : dA[3]));
					const int db = pbq == 0 ? dA[0] : (pbq == 1 ? dA[1] : (pbq == 2 ? dA[2] : dA[3]));
					const double yl = row_pick_f64(ye, pe & 15), ylb = row_pick_f64(yb, pb & 15);
					const int dl = row_pick_i32(de, pe & 15), dlb = row_pick_i32(db, pb & 15);
					y2 = pe > 0 ? ylb : y1_in;
					y1 = yl;
					fd2 = pe > 0 ? dlb : fd1;
					fd1 = dl;
				}
				if (meta & kWhbRecLock) {  // the decoder locked on this sample: the average it froze (whb.cpp:653-654)
					const int delta = (int)y1 - (int)rec.w;
					bad = bad || delta > tol || delta < -tol || (delta != 0 && (meta & kWhbRecAmb));
					carry = (meta & kWhbRecClosed) ? 0 : delta;
				}
			}
			// ---- the rings move on
#pragma unroll
			for (int k = 0; k < kVerRecAhead; k++)
				R[k] = R[k + 1];
			R[kVerRecAhead] = rnew;
#pragma unroll
			for (int k = 0; k < kVerAhead; k++)
#pragma unroll
				for (int q = 0; q < 4; q++)
					D[k][q] = D[k + 1][q];
#pragma unroll
			for (int q = 0; q < 4; q++)
				D[kVerAhead][q] = dnew[q];
			v++;
		}
	}
	if (active && li == 0) {
		st.carry = carry_in;
		st.pad_ = 0;
		T.whbx0[s] = st;  // the exact state this submit started from, and the carry (a redo needs both)
		// a stream whose speculative pass started from a state that a redo has replaced since is redone as well
		bad = bad || T.whbseen[s] != T.whbgen[s];
		if (whb_hook_force_fail(T) > 0 && (s + T.whb_submit_seq) % whb_hook_force_fail(T) == 0)
			bad = true;  // tests
		st.y1 = y1;
		st.y2 = y2;
		st.fd1 = fd1;
		st.fd2 = fd2;
		st.carry = st.pad_ = 0;
		T.whbx[s] = st;
		carry_io[s] = bad ? 0 : carry;  // (the exact kernel freezes the exact average: nothing to carry)
		T.whbfail[s] = bad ? 1 : 0;
	}
}

#include "whb_check.h"

// ------------------------------------------------------------------------------------------------ K5
// decoder::store_bit / flush for TFA_1 and the TFA_2 family, in two stages:
//   K5a decode_kernel  lane per WINDOW: every window of a chain ends with decoder::flush, which re-arms the decoder
//                      (sr_cnt = -1, byte_cnt = 0; tfa1.cpp:115-117, tfa2.cpp:213-216/276-278), so the bits of one
//                      window can be decoded without the windows before it.  What does cross windows: TFA_1's
//                      shift register (not cleared by flush) -- re-created from the tail of the preceding windows'
//                      bits -- and the stale bytes of rdata[] beyond this window's byte_cnt, handled in K5b.
//   K5b commit_kernel  lane per (stream, slot) for TFA_1, commit_wave_kernel wave per (stream, slot) for the TFA_2 family: walks the windows in order: validates/repairs the tfa2
//                      last_bit_idx speculation, overlays the windows' rdata bytes in order (rdata persistence),
//                      emits the flush events and commits ChainState for the next submit.  O(64 bytes) per window.
__device__ __forceinline__ const uint32_t *win_bits(const WinTables &T, int c, int j, int og)
{
	return T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j;
}

// TFA_1: the decoder's shift register at the start of window j (> 0) = the last 32 bits handed to store_bit
// before it (tfa1.cpp:122: sr = (sr >> 1) | (bit << 31), bits are stored LSB first: same order)
__device__ __forceinline__ uint32_t tfa1_sr_before(const WinTables &T, const ChainState &st, int c, int j)
{
	uint32_t sr = 0;
	int have = 0;  // bits gathered (the newest at the top of sr)
	for (int k = j - 1; k >= 0 && have < 32; k--) {
		const int nb = T.result[(size_t)c * T.cap + k].nbits;
		if (nb <= 0)
			continue;
		const uint32_t *bits = win_bits(T, c, k, T.open[(size_t)c * T.cap + k]);
		const int take = nb < 32 - have ? nb : 32 - have;  // the last `take` bits of window k
		const int p0 = nb - take;
		const uint32_t lo = bits[p0 >> 5], hi = ((p0 + take - 1) >> 5) != (p0 >> 5) ? bits[(p0 >> 5) + 1] : 0u;
		const unsigned long long w = ((unsigned long long)hi << 32) | lo;
		const uint32_t piece = (uint32_t)(w >> (p0 & 31)) & (take >= 32 ? ~0u : (1u << take) - 1u);
		// these bits are OLDER than what is gathered so far: they go below
		sr = (have == 0) ? (take >= 32 ? piece : piece << (32 - take))
				 : (sr | (piece << (32 - have - take)));
		have += take;
	}
	if (have < 32)
		sr |= have ? (st.sr >> have) : st.sr;
	return sr;
}

template <int KIND>
__device__ __forceinline__ void decode_window(int c, int j, int n_streams, const ChainLaunch &L, const WinTables &T,
					      uint8_t *__restrict__ my_rdata)
{
	const int a = c / n_streams, s = c - a * n_streams;
	const ChainState &st = L.states[a][s];
	const WinResult r = T.result[(size_t)c * T.cap + j];
	const int og = T.open[(size_t)c * T.cap + j];
	const uint32_t *bits = win_bits(T, c, j, og);
	Dec d{ 0u, -1, 0, 0, 0, 0, 0, 0, 0, 0u, 0u, my_rdata };
	if (j == 0) {  // the chain's first window of this submit continues from the carried decoder state
		const uint4 *src = reinterpret_cast<const uint4 *>(st.rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(my_rdata);
#pragma unroll
		for (int q = 0; q < 4; q++)
			dst[q] = src[q];
		d.sr = st.sr;
		d.sr_cnt = st.sr_cnt;
		d.byte_cnt = st.byte_cnt;
		d.invert = st.invert;
	} else if (KIND == 0) {
		d.sr = tfa1_sr_before(T, st, c, j);
	}
	int maxlen = d.byte_cnt;
	const int nbits = r.nbits;
	uint32_t wnext = nbits > 0 ? bits[0] : 0u;
	for (int n = 0; n < nbits; n += 32) {
		const uint32_t wbits = wnext;
		if (n + 32 < nbits)
			wnext = bits[(n >> 5) + 1];  // next word in flight while this one is decoded
		const int cnt = nbits - n < 32 ? nbits - n : 32;
		for (int q = 0; q < cnt; q++) {
			store_bit<KIND>(d, (wbits >> q) & 1);  // decoder::store_bit
			maxlen = d.byte_cnt > maxlen ? d.byte_cnt : maxlen;
		}
	}
	WinDecode &o = T.decode[(size_t)c * T.cap + j];
	o.sr = d.sr;
	o.sr_cnt = d.sr_cnt;
	o.byte_cnt = d.byte_cnt;
	o.invert = d.invert;
	o.wlen = j == 0 ? 64 : (maxlen < 64 ? maxlen : 64);
	const uint4 *src = reinterpret_cast<const uint4 *>(my_rdata);
	uint4 *dst = reinterpret_cast<uint4 *>(o.vals);
#pragma unroll
	for (int q = 0; q < 4; q++)
		dst[q] = src[q];
}

__global__ __launch_bounds__(64) void decode_kernel(int n_streams, ChainLaunch L, WinTables T, int kind)
{
	__builtin_amdgcn_s_setprio(3);  // see windows_kernel
	__shared__ __attribute__((aligned(16))) uint8_t rdata_lds[64 * 256];
	uint8_t *my_rdata = rdata_lds + 256 * threadIdx.x;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	const uint32_t tid = blockIdx.x * 64 + threadIdx.x, nthreads = gridDim.x * 64;
	for (int q = 2 * kind; q < 2 * kind + 2; q++) {  // long windows first
		const uint32_t count = T.queue[q].count;
		for (uint32_t idx = tid; idx < count; idx += nthreads) {
			const uint2 it = T.items[(size_t)q * total + idx];
			if (kind == 0)
				decode_window<0>((int)it.x, (int)it.y, n_streams, L, T, my_rdata);
			else
				decode_window<1>((int)it.x, (int)it.y, n_streams, L, T, my_rdata);
		}
	}
}

// WAVE: the whole wave walks ONE chain in lock step (every lane computes the same); only lane 0 reports events and
// stores state.  That way the rare exact re-slice of a window is the wave-cooperative slicer, not one lane's.
template <int KIND, bool WAVE>
__device__ __forceinline__ void commit_body(int a, int s, int n_streams, int n_blocks, long long sample_base,
					    const uint32_t *__restrict__ dec, size_t dec_stride,
					    const int16_t *__restrict__ ld16, const ChainLaunch &L, const WinTables &T,
					    tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags,
					    uint8_t *__restrict__ my_rdata)
{
	const int M = n_blocks * kBlockDec;
	const ChainParams &p = L.params[a];
	ChainState &st = L.states[a][s];
	const int c = a * n_streams + s;
	const int count = T.count[c];
	const bool lead = !WAVE || threadIdx.x == 0;
	EmitCtx e{ events, eb, flags, (uint32_t)s, L.slot[a], p.sensor_type, sample_base, !lead };
	{  // rdata[0 .. 64) as the previous submit left them (only these are ever looked at: INTEGRATION.md)
		const uint4 *src = reinterpret_cast<const uint4 *>(st.rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(my_rdata);
#pragma unroll
		for (int q = 0; q < 4; q++)
			dst[q] = src[q];
	}
	Dec d{ st.sr, st.sr_cnt, st.byte_cnt, st.invert, st.synced, st.w_last_bit, st.psk, st.last_psk, st.nrzs, st.lfsr, st.seq,
	       my_rdata };
	if (KIND == 1 && !WAVE) {
		// First only the edge-timing check of every window (see below).  A chain with a window that fails it is handed
		// to commit_wave_kernel, where the exact re-slice is the wave-cooperative slicer; nothing of it is committed here.
		int lbi = st.last_bit_idx, lbi_block = -1;
		bool ok = true;
		for (int j = 0; j < count && ok; j++) {
			const int close = T.close[(size_t)c * T.cap + j];
			const int last = close < M ? close : M - 1;
			const WinResult *rr = &T.result[(size_t)c * T.cap + j];
			if (j > 0 && rr->first_cand_g >= 0) {
				const int index_c = 2 * (rr->first_cand_g & (kBlockDec - 1));
				const int lbi_c = rebase_lbi(lbi, lbi_block, rr->first_cand_g >> 13);
				const int tdiff = index_c - lbi_c;
				ok = (index_c > lbi_c + 8) && !(tdiff > p.spb / 4 && tdiff < 32 * p.spb);
			}
			lbi = (j == 0 || rr->first_cand_g >= 0) ? rr->lbi_out : rebase_lbi(lbi, lbi_block, last >> 13);
			lbi_block = last >> 13;
		}
		if (!ok) {
			const size_t total = (size_t)L.n_active * n_streams * T.cap;
			const uint32_t idx = atomicAdd(&T.queue[kDeferQueue].count, 1u);
			T.items[(size_t)kNQueues * total + idx] = make_uint2((uint32_t)a, (uint32_t)s);
			return;
		}
	}
	int lbi = st.last_bit_idx;  // true last_bit_idx, relative to lbi_block
	int lbi_block = -1;
	const WinResult *last_r = nullptr;
	for (int j = 0; j < count; j++) {
		const int og = T.open[(size_t)c * T.cap + j];
		const int close = T.close[(size_t)c * T.cap + j];
		const int last = close < M ? close : M - 1;
		WinResult *rr = &T.result[(size_t)c * T.cap + j];
		if (KIND == 1) {
			if (j > 0) {
				// window j was sliced assuming last_bit_idx far in the past (kSpecLbi); check with the true value
				if (rr->first_cand_g >= 0) {
					const int bc = rr->first_cand_g >> 13;
					const int index_c = 2 * (rr->first_cand_g & (kBlockDec - 1));
					const int lbi_c = rebase_lbi(lbi, lbi_block, bc);
					const int tdiff = index_c - lbi_c;
					// the speculative run saw: glitch test passed, edge counted, nothing emitted, last_bit kept
					// (tfa2.cpp:391-409 with a huge tdiff).  The true run does the same iff:
					const bool same = (index_c > lbi_c + 8) && !(tdiff > p.spb / 4 && tdiff < 32 * p.spb);
					if (WAVE && !same) {  // slice and decode this window again, exactly (rare; the lane-per-chain
							      // form never gets here: it deferred the chain above)
						if (lead)
							atomicAdd(&T.stats[3], 1ull);
						GroupStats unused = { 0, 0 };
						coop_tfa2(c, j, n_streams, M, dec, dec_stride, ld16, L, T, nullptr, unused, true,
							  rebase_lbi(lbi, lbi_block, og >> 13));
						__threadfence();  // lane 0's stores (bits, result) before every lane reads them
						__syncthreads();
						uint4 keep[4];
#pragma unroll
						for (int q = 0; q < 4; q++)
							keep[q] = reinterpret_cast<uint4 *>(my_rdata)[q];
						decode_window<1>(c, j, n_streams, L, T, my_rdata);
#pragma unroll
						for (int q = 0; q < 4; q++)
							reinterpret_cast<uint4 *>(my_rdata)[q] = keep[q];
					}
					lbi = rr->lbi_out;
				} else {
					lbi = rebase_lbi(lbi, lbi_block, last >> 13);  // no candidate edge: it just ages
				}
			} else {
				lbi = rr->lbi_out;  // window 0 always runs with the exact carried value
			}
			lbi_block = last >> 13;
		}
		// the window's rdata writes on top of what was there
		const WinDecode *wd = &T.decode[(size_t)c * T.cap + j];
		const int wl = wd->wlen;
		const uint32_t *vsrc = reinterpret_cast<const uint32_t *>(wd->vals);
		uint32_t *vdst = reinterpret_cast<uint32_t *>(my_rdata);
		for (int b = 0; b < wl; b += 4) {
			const uint32_t v = vsrc[b >> 2];
			if (wl - b >= 4)
				vdst[b >> 2] = v;
			else {
				const uint32_t m = (1u << (8 * (wl - b))) - 1u;
				vdst[b >> 2] = (vdst[b >> 2] & ~m) | (v & m);
			}
		}
		d.sr = wd->sr;
		d.sr_cnt = wd->sr_cnt;
		d.byte_cnt = wd->byte_cnt;
		d.invert = wd->invert;
		if ((flags & TFREC_AMD_F_BITS) && rr->nbits > 0)  // parity mode: what the slicer handed to store_bit in this window
			emit_bits(e, d.seq, og, 0, win_bits(T, c, j, og), rr->nbits);
		if (rr->closed)  // the window's timeout fired: decoder::flush
			flush<KIND>(e, d, rr->rssi_i, KIND == 1 ? rr->offset : 0, last);
		last_r = rr;
	}
	// ---- commit the state the next submit starts from
	if (!lead)
		return;
	const bool open_at_end = last_r && !last_r->closed;
	if (open_at_end) {
		st.mark_lvl = last_r->mark_lvl;
		st.rssi_i = last_r->rssi_i;
		st.bitcnt = last_r->bitcnt;
		st.dmin = last_r->dmin;
		st.dmax = last_r->dmax;
		st.offset = last_r->offset;
		st.last_bit = last_r->last_bit;
	} else {
		st.mark_lvl = 0;
		st.rssi_i = 0;
		st.bitcnt = 0;
		st.dmin = 32767;
		st.dmax = -32767;
		st.offset = 0;
		st.last_bit = 0;
	}
	if (KIND == 0)
		st.last_bit_idx = open_at_end ? rebase_lbi(last_r->lbi_out, (M - 1) >> 13, n_blocks - 1) : 0;
	else
		st.last_bit_idx = rebase_lbi(lbi, lbi_block, n_blocks - 1);
	st.timeout_cnt = T.timeout_next[c];
	{
		const uint32_t lw = dec[(size_t)s * dec_stride + M - 1];
		st.prev_i = (int)(int16_t)(lw & 0xffff);
		st.prev_q = (int)lw >> 16;
	}
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(my_rdata);
		uint4 *dst = reinterpret_cast<uint4 *>(st.rdata);
#pragma unroll
		for (int q = 0; q < 4; q++)
			dst[q] = src[q];
	}
	st.sr = d.sr;
	st.sr_cnt = d.sr_cnt;
	st.byte_cnt = d.byte_cnt;
	st.invert = d.invert;
	st.synced = d.synced;
	st.seq = d.seq;
}

__global__ __launch_bounds__(64) void commit_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						    const int16_t *__restrict__ ld16, int n_streams, int n_blocks,
						    long long sample_base, ChainLaunch L, WinTables T,
						    tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags,
						    int lanes, int want_kind)
{
	__builtin_amdgcn_s_setprio(3);  // see windows_kernel
	__shared__ __attribute__((aligned(16))) uint8_t rdata_lds[64 * 256];
	uint8_t *my_rdata = rdata_lds + 256 * threadIdx.x;
	const int a = blockIdx.y;
	const int s = blockIdx.x * lanes + threadIdx.x;
	if ((int)threadIdx.x >= lanes || s >= n_streams)
		return;
	const int kind = L.params[a].kind;
	if (kind != want_kind)
		return;
	if (kind == 0)
		commit_body<0, false>(a, s, n_streams, n_blocks, sample_base, dec, dec_stride, ld16, L, T, events, eb, flags, my_rdata);
	else if (kind == 1)
		commit_body<1, false>(a, s, n_streams, n_blocks, sample_base, dec, dec_stride, ld16, L, T, events, eb, flags, my_rdata);
}

// TFA_2 family, the chains commit_kernel deferred: one wave per chain
__global__ __launch_bounds__(64) void commit_wave_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
							 const int16_t *__restrict__ ld16, int n_streams, int n_blocks,
							 long long sample_base, ChainLaunch L, WinTables T,
							 tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb,
							 uint32_t flags)
{
	__builtin_amdgcn_s_setprio(3);  // see windows_kernel
	__shared__ __attribute__((aligned(16))) uint8_t rdata_lds[256];
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	const uint32_t count = T.queue[kDeferQueue].count;
	for (uint32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
		const uint2 it = T.items[(size_t)kNQueues * total + idx];
		commit_body<1, true>((int)it.x, (int)it.y, n_streams, n_blocks, sample_base, dec, dec_stride, ld16, L, T, events, eb,
				     flags, rdata_lds);
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ launch
hipError_t launch_fmdev(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			size_t mask_stride, const uint32_t *prevdec, int16_t *fmdev, size_t fmdev_stride, EventBuf *eb,
			int n_streams, int n_blocks, int wmax, double flag_eps);

hipError_t launch_pipeline(const PipeCtl &P, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			   size_t mask_stride, const int16_t *fmdev, size_t fmdev_stride, int n_streams, int n_blocks,
			   long long sample_base, const ChainLaunch &L, const WinTables &T, int16_t *ld16, int32_t *dev32,
			   tfrec_amd_event *events, EventBuf *eb, uint32_t flags)
{
	// P.tev (optional, kTimingMarks events), one interval per kernel:
	//   ws : 0 | windows | 21
	//   k2 : 1 | spec | 2 | repair | 3 | fix | 4(k2)        cs : 23 | slicer | 5 | coop_slicer | 6 | decode | 7 | commit | 8
	//   kw : 9 | spec | 10 | repair | 11 | fix | 12          aux: 22 | whb_demod (+ decoder tail) | 13 = 14 = 15
	//   k2 : 24 | fmdev | 25  (only when the discriminator pass runs here)
	//   t1 : 16 | mark + slicer | 17 | coop_slicer | 18 | decode | 19 | commit | 20
	//   vx : 26 | whb_verify | 27          cz : 28 | coop_slicer (TFA_2 family, when split off cs) | 29
	auto mark = [&](int k, hipStream_t s_) {
		if (P.tev)
			(void)hipEventRecord(P.tev[k], s_);
	};
	// WHAT-IF experiments only (results are wrong): leave kernels out to see what each costs the batch period
	static const int skip = TFREC_KNOB_INT("SKIP", 0, 0, 1 << 16);
	hipError_t e = hipSuccess;
#define TRY(x)                          \
	do {                            \
		if ((e = (x)) != hipSuccess) \
			return e;       \
	} while (0)
	if (L.n_active == 0) {
		for (int k = 0; k < 3; k++)
			TRY(hipEventRecord(P.done[k], k == 0 ? P.cs : (k == 1 ? P.aux : P.t1)));
		return hipSuccess;
	}
	// Lanes per wave for the serial kernels (tunable for experiments: TFREC_AMD_LANES_*).  Measured on MI355X:
	// fewer lanes per wave (less lock-step divergence, more waves) is NOT faster -- full waves win.
	static const int lanes_chain = TFREC_KNOB_INT("LANES_CHAIN", 64, 1, 64), lanes_win = TFREC_KNOB_INT("LANES_WIN", 64, 1, 64);
	dim3 block(64);
	dim3 grid((n_streams + lanes_chain - 1) / lanes_chain, L.n_active);
	const int win_blocks = std::min(16384, (int)(((size_t)n_streams * n_blocks * 2 + lanes_win - 1) / lanes_win));
	// biquad segments: at most (M/32 + windows)/kSegSlots + 1 per chain
	const int seg_blocks = std::min(16384, (int)(((size_t)L.n_active * n_streams * ((size_t)n_blocks * (kBlockDec / 32) / kSegSlots + 4) +
						       lanes_win - 1) / lanes_win));
	// The repair passes run ~35 slots per segment on average, and the whole segment (116) for the few whose trajectories
	// never meet: with a lane per segment a wave is as slow as its slowest lane and two thirds of its lanes idle.
	// Several segments per lane instead (the flat loop of spec_biquad_kernel hands a lane the next one): a
	// eighth of the waves; at least 256 so that small batches keep their parallelism.
	static const int repair_div = TFREC_KNOB_INT("REPAIR_DIV", 8, 1, 64);  // (256-slot segments: 6-8 measured equal; round 4 had 12 at 128 slots)
	const int repair_blocks = std::min(seg_blocks, std::max(256, seg_blocks / repair_div));
	// The speculative pass with a fifth of the worst-case waves (~830 at 1024 streams: 1-2 segments of 256 slots per lane).  A lane
	// reads 64 (+4) bytes per slot at an arbitrary 2-byte offset of its row, so consecutive slots share a 128-byte line;
	// with a lane per segment the lines in flight (2540 waves x 64 lanes x 2 lines = 40 MB) never survived in the 32 MB
	// of L2 until the lane came back: the pass fetched 2.9 GB for 1.1 GB of input.  With ~1000 waves: 1.4 GB, and the
	// batch 2 % shorter.  (A sixteenth starves the WHB chain.)
	static const size_t lds_pad_spec = (size_t)TFREC_KNOB_INT("LDS_PAD_SPEC", 0, 0, 48 << 10);
	static const int spec_div = TFREC_KNOB_INT("SPEC_DIV", 5, 1, 64);  // (256-slot segments: 4-6 measured equal, 3 and 8 worse; round 4 had 8 at 128 slots)
	const int spec_blocks = std::min(seg_blocks, std::max(256, seg_blocks / spec_div));
	// (few chains: the lanes of the lane-per-window kernels are mostly idle anyway and latency is all that counts)
	static const int long_window_env = TFREC_KNOB_INT("COOP_MIN", 0, 0, 1 << 30);
	const int long_window = long_window_env >= 356 ? long_window_env
						       : ((size_t)n_streams * L.n_active >= 1024 ? kLongWindow : kLongWindow / 2);
	// long windows: at most M / long_window per chain
	const int coop_blocks = std::min(TFREC_KNOB_INT("COOP_BLOCKS", 32768, 1, 1 << 20), std::max(1, (int)std::min<size_t>((size_t)L.n_active * n_streams *
								((size_t)n_blocks * kBlockDec / (size_t)std::max(long_window, 356) + 1), 1u << 30)));
	const int dec_blocks = std::min(16384, std::max(1, win_blocks));
	bool has_whb = false, has_tfa2 = false, has_tfa1 = false;
	for (int a = 0; a < L.n_active; a++) {
		has_whb = has_whb || L.params[a].kind == 2;
		has_tfa2 = has_tfa2 || L.params[a].kind == 1;
		has_tfa1 = has_tfa1 || L.params[a].kind == 0;
	}
	// ---- window scan: behind the front end on its stream, or -- deep layout -- at the head of the WHB biquad stream
	// (the front-end stream is the busiest of all: 0.3-1.1 ms less on it per batch); consecutive scans stay in order
	// on one stream either way (timeout_carry)
	if (P.ws != P.fs)
		TRY(hipStreamWaitEvent(P.ws, P.ev_front, 0));
	TRY(hipMemsetAsync(T.queue, 0, (kNQueues + 1) * sizeof(WorkQueue), P.ws));
	mark(0, P.ws);
	hipLaunchKernelGGL(windows_kernel, dim3(n_streams), block, 0, P.ws, mask, mask_stride, n_streams, n_blocks, L, T,
			   long_window);
	mark(21, P.ws);
	TRY(hipEventRecord(P.ev_win, P.ws));
	// Independent kernel chains after the scan (they touch disjoint state):
	//   kw -> aux: WHB          spec -> repair -> fix (biquad) | whb_demod -> whb_decode -> whb_commit
	//   k2 -> cs : TFA_2 family spec -> repair -> fix (biquads) | slicer -> coop_slicer -> decode -> commit
	//   t1       : TFA_1        mark -> slicer -> coop_slicer -> decode -> commit
	// ---- WHB
	int whb_verify = -1;  // the WHB slot, when its speculative stage 2 ran
	// The discriminator pass (only the TFA_2 family reads its output) at the head of kw instead of k2: with the WHB stage 2
	// speculated, k2 (discriminator + three biquad passes + verify) was the longest stream of the batch and kw half idle
	static const int fmdev_kw = TFREC_KNOB_INT("FMDEV_KW", 0, 0, 1);  // (measured: 8.2 instead of 7.3 ms per batch -- the pass stretches to 5 ms there)
	const bool fm_on_kw = fmdev_kw && has_whb && has_tfa2 && P.fmdev_wmax > 0 && P.kw != P.k2;
	if (has_whb) {
		TRY(hipStreamWaitEvent(P.kw, P.ev_win, 0));
		if (fm_on_kw) {
			mark(24, P.kw);
			TRY(launch_fmdev(P.kw, dec, dec_stride, mask, mask_stride, P.prevdec, P.fmdev_out, fmdev_stride, eb, n_streams,
					 n_blocks, P.fmdev_wmax, P.fm_flag_eps));
			mark(25, P.kw);
			TRY(hipEventRecord(P.ev_fm, P.kw));
		}
		mark(9, P.kw);
		if (!(skip & 256))
		hipLaunchKernelGGL((spec_biquad_kernel<true, 0>), dim3(spec_blocks), block, K3Tile<true>::kSize + lds_pad_spec, P.kw, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		mark(10, P.kw);
		if (!(skip & 256))
		hipLaunchKernelGGL((spec_biquad_kernel<true, 1>), dim3(repair_blocks), block, K3Tile<true>::kSize + lds_pad_spec, P.kw, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		if (!(skip & 256))
		hipLaunchKernelGGL((spec_biquad_kernel<true, 2>), dim3(repair_blocks), block, K3Tile<true>::kSize + lds_pad_spec, P.kw, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		mark(11, P.kw);
		hipLaunchKernelGGL(fix_biquad_kernel, dim3(n_streams, L.n_active), block, 0, P.kw, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, 2);
		mark(12, P.kw);
		TRY(hipEventRecord(P.ev_kw, P.kw));
		TRY(hipStreamWaitEvent(P.aux, P.ev_kw, 0));
		mark(22, P.aux);
		for (int a = 0; a < L.n_active; a++)
			if (L.params[a].kind == 2) {
				// 4 KB of dynamic LDS for the decoder tail (64 lanes x rdata[0 .. 64)).  The kernel is launched while the other
				// chains' kernels occupy the chip, and its one-wave workgroups go wherever LDS is free: beside six resident
				// front-end workgroups (25 KB each of the CU's 160 KB) the 17 KB it used to ask for did not fit at all, so it
				// trickled onto the chip at the front end's pace.  TFREC_AMD_WHB_LDS raises it (caps the workgroups per CU).
				static const int whb_lds = std::max(64 * 64, TFREC_KNOB_INT("WHB_LDS", 0, 0, 64 << 10));
				const dim3 wgrid(n_streams), wblock(64);
				const int wlds = whb_lds;
				// TFREC_AMD_WHB_EXACT=1: the wave-per-stream recurrence (exact by itself: no verification pass).  BITS mode
				// (parity / debug) uses it too.
				static const int whb_exact = TFREC_KNOB_INT("WHB_EXACT", 0, 0, 1);
				if (whb_exact || (flags & TFREC_AMD_F_BITS)) {
					hipLaunchKernelGGL((whb_demod_kernel<true, false>), wgrid, wblock, wlds, P.aux, dec, dec_stride, dev32, n_streams,
							   n_blocks, sample_base, L, a, T, events, eb, flags);
				} else {
					if (!(skip & 32))
					hipLaunchKernelGGL((whb_demod_kernel<false, false>), wgrid, wblock, wlds, P.aux, dec, dec_stride, dev32, n_streams,
							   n_blocks, sample_base, L, a, T, events, eb, flags);
					whb_verify = a;
				}
				mark(13, P.aux);
				mark(14, P.aux);
				mark(15, P.aux);
			}
	}
	if (whb_verify >= 0) {  // stage C of WHB: a serial chain per lane, a few dozen waves
		if (P.vx != P.aux) {
			TRY(hipEventRecord(P.ev_aux, P.aux));
			TRY(hipStreamWaitEvent(P.vx, P.ev_aux, 0));
		}
		mark(26, P.vx);
		// The check: the exact chain a stream per LANE over the filter's input sequence, then the records against it (whb_check.h);
		// TFREC_AMD_WHB_CHECK_ROWS=1: rounds 3-5's kernel, a stream per row of 16 lanes
		static const int check_rows = TFREC_KNOB_INT("WHB_CHECK_ROWS", 0, 0, 1);
		if (check_rows) {
			if (!(skip & 16))
			hipLaunchKernelGGL(whb_verify_kernel, dim3((n_streams + 15) / 16), dim3(256), 0, P.vx, dev32, n_streams, n_blocks, L, whb_verify,
					   T, P.whb_carry);
		} else {
			static const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&whb_chain_kernel),
									     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChkLdsBytes);
			if (lds_ok != hipSuccess)
				return lds_ok;
			if (!(skip & 16)) {
			hipLaunchKernelGGL(whb_chain_kernel, dim3((n_streams + kChkStreams - 1) / kChkStreams), dim3(64 * (1 + kChkProducers)), kChkLdsBytes,
					   P.vx, n_streams, L, whb_verify, T);
			hipLaunchKernelGGL(whb_check_kernel, dim3(n_streams), block, 0, P.vx, n_streams, L, whb_verify, T, P.whb_carry);
			}
		}
		// ... and the streams it failed (normally none: every workgroup returns at once) again, exactly -- on the private
		// state array: the speculative kernels of the submits behind this one work in place on the live one meanwhile
		ChainLaunch Lr = L;
		WinTables Tr = T;
		Tr.whbpub = L.states[whb_verify];
		Lr.states[whb_verify] = T.whbscr;
		hipLaunchKernelGGL((whb_demod_kernel<true, true>), dim3((n_streams + 63) / 64), block, 64 * 64, P.vx, dec, dec_stride, dev32, n_streams,
				   n_blocks, sample_base, Lr, whb_verify, Tr, events, eb, flags);
		mark(27, P.vx);
		TRY(hipEventRecord(P.done[1], P.vx));
	} else {
		TRY(hipEventRecord(P.done[1], P.aux));
	}
	static const int head_chunks = std::max(1, TFREC_KNOB_INT("HEAD_CHUNKS", 64, 0, 1 << 30));
	// the slicer -> decoder chain of one protocol kind (0: TFA_1, 1: TFA_2 family) on stream s_
	auto slicer_chain = [&](int kind, hipStream_t s_, int m0) {
		if (kind == 0)
			if (!(skip & 8))
			hipLaunchKernelGGL(mark_kernel, dim3(std::max(1, win_blocks / 16)), dim3(256), 0, s_, dec, dec_stride, n_streams,
					   n_blocks, L, T);
		// The lanes take their windows from a queue, so the wave count is a free parameter: fewer waves = fewer registers
		// held for milliseconds by a latency-bound kernel (the front end beside it lives on what is left), more windows per lane
		static const int slicer_div = TFREC_KNOB_INT("SLICER_DIV", 1, 1, 64);
		// (TFREC_AMD_LDS_PAD_SLICER / _SPEC: sensitivity experiments -- extra dynamic LDS bytes per workgroup of the lane-per-window
		// slicers / the biquad passes: how much of the period is these kernels' LDS footprint beside the front end's 16.6 KB tiles)
		static const size_t lds_pad_slicer = (size_t)TFREC_KNOB_INT("LDS_PAD_SLICER", 0, 0, 48 << 10);
		const size_t slds = (kind == 0 ? 8 : 4) * 64 * sizeof(uint4) + lds_pad_slicer;
		const bool split = kind == 1 && P.cz != nullptr;
		if (split) {
			// the long windows' heads first (few windows: a small grid), then -- beside each other -- their tails on cz and
			// the short windows here: stage B of the TFA_2 family was the longest chain of the batch (slicers 3.5 ms +
			// cooperative slicers 2.6 ms, one after the other)
			hipLaunchKernelGGL(slicer_kernel, dim3(std::max(64, win_blocks / 8)), block, slds, s_, dec, dec_stride, ld16, n_streams,
					   n_blocks, L, T, lanes_win, head_chunks, kind, 1);
			(void)hipEventRecord(P.ev_heads, s_);
			(void)hipStreamWaitEvent(P.cz, P.ev_heads, 0);
			mark(28, P.cz);
			hipLaunchKernelGGL(coop_slicer_kernel, dim3(coop_blocks), block, 0, P.cz, dec, dec_stride, ld16, n_streams, n_blocks, L,
					   T, kind);
			mark(29, P.cz);
			(void)hipEventRecord(P.ev_coop, P.cz);
			hipLaunchKernelGGL(slicer_kernel, dim3(std::max(64, win_blocks / slicer_div)), block, slds, s_, dec, dec_stride, ld16,
					   n_streams, n_blocks, L, T, lanes_win, head_chunks, kind, 2);
			mark(m0 + 1, s_);
			(void)hipStreamWaitEvent(s_, P.ev_coop, 0);
		} else {
			if (!(skip & (kind == 0 ? 8 : 4)))
			hipLaunchKernelGGL(slicer_kernel, dim3(std::max(64, win_blocks / slicer_div)), block, slds, s_, dec, dec_stride, ld16, n_streams, n_blocks, L, T,
					   lanes_win, head_chunks, kind, 0);
			mark(m0 + 1, s_);
			if (!(skip & (kind == 0 ? 2 : 1)))
			hipLaunchKernelGGL(coop_slicer_kernel, dim3(coop_blocks), block, 0, s_, dec, dec_stride, ld16, n_streams, n_blocks, L,
					   T, kind);
		}
		mark(m0 + 2, s_);
		hipLaunchKernelGGL(decode_kernel, dim3(dec_blocks), block, 0, s_, n_streams, L, T, kind);
		mark(m0 + 3, s_);
		hipLaunchKernelGGL(commit_kernel, grid, block, 0, s_, dec, dec_stride, ld16, n_streams, n_blocks, sample_base, L, T,
				   events, eb, flags, lanes_chain, kind);
		if (kind == 1)  // the few chains (normally none) with a window to slice again
			hipLaunchKernelGGL(commit_wave_kernel, dim3(256), block, 0, s_, dec, dec_stride, ld16, n_streams, n_blocks,
					   sample_base, L, T, events, eb, flags);
		mark(m0 + 4, s_);
	};
	// ---- TFA_2 family
	bool t1_waits = false;
	if (has_tfa2) {
		TRY(hipStreamWaitEvent(P.k2, P.ev_win, 0));
		if (fm_on_kw) {
			TRY(hipStreamWaitEvent(P.k2, P.ev_fm, 0));
		} else if (P.fq && P.fmdev_wmax > 0) {
			// (TFREC_AMD_FMDEV_OWN) the discriminator pass on a stream of its own: it needs the front end only, not the window
			// scan, and k2 -- discriminator + five biquad kernels -- is the stream that sets the period
			TRY(hipStreamWaitEvent(P.fq, P.ev_front, 0));
			mark(24, P.fq);
			TRY(launch_fmdev(P.fq, dec, dec_stride, mask, mask_stride, P.prevdec, P.fmdev_out, fmdev_stride, eb, n_streams,
					 n_blocks, P.fmdev_wmax, P.fm_flag_eps));
			mark(25, P.fq);
			TRY(hipEventRecord(P.ev_fm, P.fq));
			TRY(hipStreamWaitEvent(P.k2, P.ev_fm, 0));
		} else if (P.fmdev_wmax > 0 && !(skip & 64)) {
			mark(24, P.k2);
			TRY(launch_fmdev(P.k2, dec, dec_stride, mask, mask_stride, P.prevdec, P.fmdev_out, fmdev_stride, eb, n_streams,
					 n_blocks, P.fmdev_wmax, P.fm_flag_eps));
			mark(25, P.k2);
		}
		// The speculative pass needs the discriminator pass and the window scan of ITS submit only (every segment starts from
		// zero): on a stream of its own (ks) it runs beside the repair passes and the chain walk of the submit before, which
		// stay on k2 -- k2 carried 5.3 ms of kernels per 5.5 ms period (spec 2.4, repairs 1.3-1.9 + 0.9, walk 0.1-0.6).
		hipStream_t sp = (P.ks && P.fq && P.fmdev_wmax > 0 && !fm_on_kw) ? P.ks : P.k2;
		if (sp != P.k2) {
			TRY(hipStreamWaitEvent(sp, P.ev_win, 0));
			TRY(hipStreamWaitEvent(sp, P.ev_fm, 0));
		}
		mark(1, sp);
		if (!(skip & 128))
		hipLaunchKernelGGL((spec_biquad_kernel<false, 0>), dim3(spec_blocks), block, K3Tile<false>::kSize + lds_pad_spec, sp, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		mark(2, sp);
		if (sp != P.k2) {
			TRY(hipEventRecord(P.ev_spec, sp));
			TRY(hipStreamWaitEvent(P.k2, P.ev_spec, 0));
		}
		if (has_tfa1 && !TFREC_KNOB_INT("T1_EARLY", 0, 0, 1 << 30)) {
			// TFA_1 needs no biquad stage and has slack: its chain starts once the speculative biquad pass (on the
			// critical path of the other chains) has had the chip to itself
			TRY(hipEventRecord(P.ev_fork, sp));
			TRY(hipStreamWaitEvent(P.t1, P.ev_fork, 0));
			t1_waits = true;
		}
		if (!(skip & 128))
		hipLaunchKernelGGL((spec_biquad_kernel<false, 1>), dim3(repair_blocks), block, K3Tile<false>::kSize + lds_pad_spec, P.k2, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		if (!(skip & 128))
		hipLaunchKernelGGL((spec_biquad_kernel<false, 2>), dim3(repair_blocks), block, K3Tile<false>::kSize + lds_pad_spec, P.k2, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, lanes_win);
		mark(3, P.k2);
		hipLaunchKernelGGL(fix_biquad_kernel, dim3(n_streams, L.n_active), block, 0, P.k2, dec, dec_stride, fmdev, fmdev_stride,
				   n_streams, n_blocks, L, T, ld16, dev32, 1);
		mark(4, P.k2);
		TRY(hipEventRecord(P.ev_k2, P.k2));
		TRY(hipStreamWaitEvent(P.cs, P.ev_k2, 0));
		mark(23, P.cs);
		slicer_chain(1, P.cs, 4);  // marks 5..8
	}
	TRY(hipEventRecord(P.done[0], P.cs));
	// ---- TFA_1
	if (has_tfa1) {
		if (!t1_waits)
			TRY(hipStreamWaitEvent(P.t1, P.ev_win, 0));
		mark(16, P.t1);
		slicer_chain(0, P.t1, 16);  // marks 17..20
	}
	TRY(hipEventRecord(P.done[2], P.t1));
#undef TRY
	return hipGetLastError();
}

}  // namespace tfrec
