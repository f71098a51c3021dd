// tfrec_amd/csrc/whb_demod.h -- K4' whb_demod_kernel: WHB stage 2, a wave per stream (speculated decision levels / the exact redo).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
			// The steps' stage-1 outputs travel through a RING of four registers with fixed roles (the loop below is unrolled by four):
			// step i reads ring[i & 3] and first loads step i + 3 into ring[(i + 3) & 3], the register of the step before.  Three
			// steps stay in flight (past the window's end: the row's next slots or the slack behind it, never used) and nothing is
			// ever MOVED: rounds 3-6 rotated cur <- nxt <- newest at the end of every step, and a move of a register whose load is
			// in flight waits for that load -- and, vmcnt counting in order, for every store of the step: a memory latency per step.
			int ring0 = wq[ln], ring1 = wq[kStep + ln], ring2 = wq[2 * kStep + ln], ring3 = 0;
			wq += 3 * kStep;  // the step the loop loads next
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
			auto step = [&](const int i, const int dev, int &fill) __attribute__((always_inline)) {
				// ---- (1) this step's inputs; the next three steps' are in flight
				const int nv = n - kStep * i < kStep ? n - kStep * i : kStep;
				const unsigned long long valid = nv < kStep ? (1ull << nv) - 1ull : ~0ull;  // the step's samples inside the window
				fill = wq[ln];
				wq += kStep;
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
#ifdef TFREC_AMD_PROFILE_WHB
				pf_tail += __builtin_readcyclecounter() - pf_mark;
#endif
			};
			for (int i = 0; i < nch; i += 4) {
				step(i, ring0, ring3);
				if (i + 1 >= nch)
					break;
				step(i + 1, ring1, ring0);
				if (i + 2 >= nch)
					break;
				step(i + 2, ring2, ring1);
				if (i + 3 >= nch)
					break;
				step(i + 3, ring3, ring2);
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
