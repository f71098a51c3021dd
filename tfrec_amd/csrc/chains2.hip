// tfrec_amd/csrc/chains2.hip -- window-parallel demodulator/decoder pipeline (the product path).
//
// Same results as the serial reference chains of chains.hip (and therefore as fsk_demod::process +
// the plugins, fm_demod.cpp:34-56, tfa1.cpp:143-190, tfa2.cpp:346-442, whb.cpp:632-707), re-cut so that
// only what the reference really serialises stays serial:
//
//   K2  windows_kernel     (windows.h, with K1b threshold_kernel) wave per stream: scan over the trigger mask -> per (stream, slot) the list of trigger
//                          windows (tfa1.cpp:147-149,179 / tfa2.cpp:351-355,428 / whb.cpp:636-641,691: a window opens
//                          at a pwr>thresh sample while the counter is 0 and its flush fires W-1 samples after the
//                          last trigger), virtual slot numbering, work queues.
//   K3  spec_biquad_kernel (biquad.h; speculate, repair, second repair) + fix_biquad_kernel (verify): the fp64 biquads
//                          (iir2::step) are the one recurrence whose state crosses windows.  TFA_2/TFA_3/TX22 read
//                          the fm_dev array of fmdev_kernel, WHB computes fm_dev_nrzs on the fly; outputs are the
//                          truncated integers the slicers consume.  Lane per 4096-sample segment (see K3 below).
//   K4  slicer_kernel      (slicers.h) lane per WINDOW (work queue, long windows first): the bit slicers of TFA_1 and the
//                          TFA_2 family are window-local state machines (state reset at window open/close) --
//                          except tfa2's last_bit_idx, which is never reset (tfa2.cpp:325-334).  It can only
//                          influence a window through its first candidate edge, so windows are run assuming a
//                          far-away last edge and the assumption is checked (and the window re-run exactly) in K5.
//                          Output: the bits handed to decoder::store_bit, packed.
//   K4a mark_kernel (slicers.h), K4b coop_slicer_kernel (coop_slicer.h): windows of 4096 samples and more: wave per window.
//   K4' whb_demod_kernel   (whb_demod.h; its tail K4'': whb_commit.h) wave per stream: WHB stage 2 (decision-level biquad, phase-change detector); the
//                          demodulator needs the decoder's has_sync() (whb.cpp:653, 677, 693), which is tracked
//                          with a lane-parallel evaluation of the (GF(2)-linear) sync search.  Output: bit runs.
//                          In its tail (K4''): whb_decoder::store_bit over the runs, lane per window, then the
//                          stream's flush events and decoder state.  <false, false>: the decision levels from a
//                          lane-parallel scan (speculated), every decision recorded;
//   K4v whb_chain_kernel + whb_check_kernel (whb_check.h): the reference's own recurrence over the filter's input sequence, a
//                          stream per LANE, then a wave per stream compares every recorded decision and carries the exact filter
//                          state; failed streams are redone by whb_demod_kernel<true, true> (the exact form), their events
//                          retracted.  whb_verify.h: the round-4 form, four streams per wave
//                          (experiments build, WHB_CHECK_ROWS=1).
//   K5  decode_kernel      (decode.h) lane per window: the decoders (store_bit) over the window's packed bits;
//       commit_kernel      lane per (stream, slot): walks the windows in order: checks the tfa2 speculation (a chain
//                          with a window to re-run goes to commit_wave_kernel, wave per chain), overlays the windows'
//                          rdata bytes, emits the flush events, commits ChainState for the next submit.
// launch_pipeline (end of this file) puts them on the context's streams: the stages of four consecutive submits run beside
// each other.
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

#include "windows.h"

#include "biquad.h"

#include "slicers.h"

#include "coop_slicer.h"

#include "whb_commit.h"

#include "whb_demod.h"

#include "whb_verify.h"

#include "whb_check.h"

#include "decode.h"

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
