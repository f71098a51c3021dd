// tfrec_amd/csrc/whb_verify.h -- K4v whb_verify_kernel: the exact chain four streams per wave (experiments: WHB_CHECK_ROWS; the product runs whb_check.h).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
