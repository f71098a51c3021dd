// tfrec_amd/csrc/whb_check.h -- WHB stage 2, the check of the speculated decisions as an exact chain PER LANE (round 6).
// Included by chains2.hip (inside namespace tfrec, behind whb_demod.h and whb_verify.h).
//
// whb_demod_kernel<false> speculates the decisions "dev < (int)avg_of" (whb.cpp:654, 662) from a lane-parallel evaluation of
// iir_avg.  The reference's own recurrence -- iir2::step in its normative association, y = ((B2 + a1 y1) + P) + a2 y2 with
// P = fma(2, t1, t0), B2 = t2, t = fl((b0 / 2) dev), dsp_dev.h: iir_step_t -- is serial per stream.  Rounds 3-5 ran it with a
// stream per ROW of 16 lanes (whb_verify_kernel: the feed-forward terms enter through DPP broadcasts): six vector instructions per
// sample for FOUR streams, 279 M of the batch's 1.95 G vector instructions.  Here a stream is a LANE:
//
//   * whb_demod_kernel<false> writes the filter's INPUT SEQUENCE of every stream (WinTables::whbdense): the filter knows neither
//     windows nor steps -- it pauses while the decoder is locked and goes on where it stopped --, so the sequence is simply every
//     stage-1 output the average ran on, in order.  Partial steps (a window's end, a lock) vanish in it;
//   * whb_chain_kernel: workgroups of one CONSUMER wave (32 streams, a lane each) and two PRODUCER waves.  The producers
//     read the sequences 64 inputs a round with coalesced loads, two rounds ahead, and lay (P, B2) pairs and the inputs
//     themselves into a double-buffered LDS image, a padded row per stream; the consumer reads its row with conflict-free
//     16-byte reads and runs the chain: five dependent fp64 operations, the conversion, the compare and one add-with-carry that
//     collects the decision bits per input -- for 32 streams at once.  Per round it stores the 64 exact decisions and the filter
//     state behind them.  It handles whole rounds only and knows nothing of windows, locks or records;
//   * whb_check_kernel (a wave per stream, throughput work): the rest of the sequence (< 64 inputs) serially, the records'
//     decisions against the exact ones (a prefix sum of the records' lengths gives each its place in the sequence), and for
//     every lock the frozen integer against (int) of the exact average there -- recomputed from the state the chain kernel left at
//     the round before, a lane per lock.  Rules, carried state and failure flags are whb_verify_kernel's.
#pragma once

#ifndef TFREC_AMD_CHK_STREAMS
#define TFREC_AMD_CHK_STREAMS 32
#endif
constexpr int kChkStreams = TFREC_AMD_CHK_STREAMS;  // streams per workgroup of whb_chain_kernel = working lanes of its consumer wave
constexpr int kChkProducers = 2;                    // producer waves
constexpr int kChkPerProducer = kChkStreams / kChkProducers;
static_assert(kChkStreams <= 64 && kChkStreams % kChkProducers == 0, "a lane per stream, the streams split evenly");
// LDS rows, in dwords.  A lane reads 16 bytes at a time from its own row: a row stride of 4 (mod 64) dwords puts the 16 lanes of
// a read phase on 16 different bank quadruples.
constexpr int kChkPairRow = 64 * 4 + 4;  // (P, B2) of 64 inputs: 16 B each
constexpr int kChkDevRow = 64 + 4;       // the inputs themselves (the compare needs the integer)
struct ChkBuf {
	uint32_t pair[kChkStreams * kChkPairRow];
	uint32_t dev[kChkStreams * kChkDevRow];
};
constexpr size_t kChkLdsBytes = 2 * sizeof(ChkBuf);

__global__ __launch_bounds__(64 * (1 + kChkProducers)) void whb_chain_kernel(int n_streams, ChainLaunch L, int a, WinTables T)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t chk_lds[];
	ChkBuf *const buf = reinterpret_cast<ChkBuf *>(chk_lds);
	const int wave = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
	const int s0 = (int)blockIdx.x * kChkStreams;
	const ChainParams &p = L.params[a];
	const double a1 = p.iir_avg.a1, a2 = p.iir_avg.a2, bh = 0.5 * p.iir_avg.b0;
	// whole rounds of the workgroup's streams (every wave computes the same numbers)
	const int my_s = s0 + ln;
	const bool my_on = ln < kChkStreams && my_s < n_streams;
	const int my_rounds = my_on ? T.whbdense_n[my_s] >> 6 : 0;
	int rmax = my_rounds;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const int other = __shfl_xor(rmax, o, 64);
		rmax = other > rmax ? other : rmax;
	}
	rmax = __builtin_amdgcn_readfirstlane(rmax);
	if (rmax == 0)
		return;
	if (wave == 0) {
		// ---------------------------------------------------------------- consumer: a stream per lane
#ifdef TFREC_AMD_CHK_CLAIM  // the whole register file of its SIMD: no other wave beside the chain (experiment)
		asm volatile("" ::: "v255", "a255");
#endif
#ifdef TFREC_AMD_CHK_PRIO  // (experiment)
		__builtin_amdgcn_s_setprio(TFREC_AMD_CHK_PRIO);
#endif
		double y1 = 0.0, y2 = 0.0;
		if (my_on) {
			const WhbExact st = T.whbx[my_s];
			y1 = st.y1;
			y2 = st.y2;
		}
		unsigned long long *const xbits = T.whbxbits + (size_t)(my_on ? my_s : 0) * T.whbx_stride;
		double2 *const xsnap = T.whbxsnap + (size_t)(my_on ? my_s : 0) * T.whbx_stride;
		__syncthreads();  // round 0 is in buf[0]
		for (int r = 0; r < rmax; r++) {
			if (r < my_rounds) {
				const ChkBuf &b = buf[r & 1];
				const double2 *prow = reinterpret_cast<const double2 *>(b.pair + ln * kChkPairRow);
				const uint4 *drow = reinterpret_cast<const uint4 *>(b.dev + ln * kChkDevRow);
				uint32_t acc[2] = { 0u, 0u };
				// eight inputs' (P, B2) pairs and integers at a time, the next eight in flight behind them
				double2 pq[2][8];
				uint4 dq[2][2];
				auto fetch = [&](int h, int g8) {
#pragma unroll
					for (int j = 0; j < 8; j++)
						pq[h][j] = prow[8 * g8 + j];
					dq[h][0] = drow[2 * g8];
					dq[h][1] = drow[2 * g8 + 1];
				};
				fetch(0, 0);
#pragma unroll
				for (int g8 = 0; g8 < 8; g8++) {
					const int h = g8 & 1;
					if (g8 + 1 < 8)
						fetch(h ^ 1, g8 + 1);
					const int dv[8] = { (int)dq[h][0].x, (int)dq[h][0].y, (int)dq[h][0].z, (int)dq[h][0].w,
							    (int)dq[h][1].x, (int)dq[h][1].y, (int)dq[h][1].z, (int)dq[h][1].w };
#pragma unroll
					for (int j = 0; j < 8; j++) {
						const double P = pq[h][j].x, B2 = pq[h][j].y;
						const double y = ((B2 + a1 * y1) + P) + a2 * y2;  // iir2::step, whb.cpp:654
						y2 = y1;
						y1 = y;
						// |0.5 dev| <= 6.6e8 and the low-pass has an L1 gain of 1.09: (int) never saturates (= x86's cvttsd2si)
						const int yi = (int)y;
						// acc = 2 acc + (dev < (int)avg): the compare's bit enters as the carry of acc + acc
#ifndef TFREC_AMD_CHK_WHATIF_NOCMP  // (timing experiment: the chain without its decisions -- results wrong)
						asm("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[g8 >> 2]) : "v"(dv[j]), "v"(yi) : "vcc");
#else
						if (j == 7) acc[g8 >> 2] += (uint32_t)yi + (uint32_t)dv[j];
#endif
					}
				}
				// (the first input of a half sits in the top bit)
				xbits[r] = (unsigned long long)__brev(acc[0]) | ((unsigned long long)__brev(acc[1]) << 32);
				xsnap[r] = make_double2(y1, y2);
			}
			__syncthreads();  // round r + 1 is in buf[(r + 1) & 1], buf[r & 1] is free
		}
	} else {
		// ---------------------------------------------------------------- producers: lane = input of the round
		const int q0 = (wave - 1) * kChkPerProducer;  // first of this wave's streams (index within the workgroup)
		// rounds, rows and the two inputs before a round's first (the filter's own input history: the last two inputs of the
		// submit before at a stream's round 0), per stream of this wave, in scalar registers
		int rounds[kChkPerProducer], h1[kChkPerProducer], h2[kChkPerProducer];
		const int32_t *row[kChkPerProducer];
#pragma unroll
		for (int q = 0; q < kChkPerProducer; q++) {
			const int s = s0 + q0 + q;
			const bool on = s < n_streams;
			rounds[q] = __builtin_amdgcn_readlane(my_rounds, (q0 + q) & 63);
			row[q] = T.whbdense + (size_t)(on ? s : 0) * T.whbdense_stride;
			const WhbExact st = T.whbx[on ? s : 0];
			h1[q] = __builtin_amdgcn_readfirstlane(st.fd1);
			h2[q] = __builtin_amdgcn_readfirstlane(st.fd2);
		}
		int cur[kChkPerProducer], nxt[kChkPerProducer];  // the lane's input of the round being laid out / of the one after it
#ifdef TFREC_AMD_CHK_PF3  // (experiment: loads three rounds ahead)
		int nx2[kChkPerProducer];
#endif
#pragma unroll
		for (int q = 0; q < kChkPerProducer; q++) {
			cur[q] = rounds[q] > 0 ? row[q][ln] : 0;
			nxt[q] = rounds[q] > 1 ? row[q][64 + ln] : 0;
#ifdef TFREC_AMD_CHK_PF3
			nx2[q] = rounds[q] > 2 ? row[q][128 + ln] : 0;
#endif
		}
		auto lay_out = [&](ChkBuf &b, int r) {  // round r from cur[] (h1, h2: the inputs 64 r - 1, 64 r - 2)
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++) {
				if (r < rounds[q]) {  // (wave-uniform)
					const int d0 = cur[q];
					const int s1 = __builtin_amdgcn_update_dpp(0, d0, 0x138, 0xf, 0xf, false);  // wave_shr:1
					const int d1 = ln == 0 ? h1[q] : s1;
					const int s2 = __builtin_amdgcn_update_dpp(0, d1, 0x138, 0xf, 0xf, false);
					const int d2 = ln == 0 ? h2[q] : s2;
					const double t0 = bh * (double)d0, t1 = bh * (double)d1, t2 = bh * (double)d2;
					const double P = __builtin_fma(2.0, t1, t0);
					*reinterpret_cast<double2 *>(b.pair + (q0 + q) * kChkPairRow + 4 * ln) = make_double2(P, t2);
					b.dev[(q0 + q) * kChkDevRow + ln] = (uint32_t)d0;
					h1[q] = __builtin_amdgcn_readlane(d0, 63);
					h2[q] = __builtin_amdgcn_readlane(d0, 62);
				}
			}
		};
		lay_out(buf[0], 0);
		__syncthreads();
		for (int r = 0; r < rmax; r++) {
			// round r + 1 is in nxt[]; the loads of round r + 2 go out before it is laid out
			int far[kChkPerProducer];
#ifdef TFREC_AMD_CHK_PF3
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++)
				far[q] = r + 3 < rounds[q] ? row[q][64 * (r + 3) + ln] : 0;
#else
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++)
				far[q] = r + 2 < rounds[q] ? row[q][64 * (r + 2) + ln] : 0;
#endif
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++)
				cur[q] = nxt[q];
			if (r + 1 < rmax)
				lay_out(buf[(r + 1) & 1], r + 1);
#ifdef TFREC_AMD_CHK_PF3
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++) {
				nxt[q] = nx2[q];
				nx2[q] = far[q];
			}
#else
#pragma unroll
			for (int q = 0; q < kChkPerProducer; q++)
				nxt[q] = far[q];
#endif
			__syncthreads();
		}
	}
}

// ---- the rest of the check: a wave per stream
constexpr int kChkLockCap = 256;  // locks of a stream collected before they are evaluated (a lane each, 64 at a time)
__global__ __launch_bounds__(64) void whb_check_kernel(int n_streams, ChainLaunch L, int a, WinTables T, int *__restrict__ carry_io)
{
	__shared__ int lk_pos[kChkLockCap], lk_avgf[kChkLockCap];
	__shared__ uint32_t lk_meta[kChkLockCap];
	const int s = (int)blockIdx.x, ln = (int)threadIdx.x;
	if (s >= n_streams)
		return;
	const ChainParams &p = L.params[a];
	const double a1 = p.iir_avg.a1, a2 = p.iir_avg.a2, bh = 0.5 * p.iir_avg.b0;
	const int N = T.whbdense_n[s], R = N >> 6;
	const int32_t *dense = T.whbdense + (size_t)s * T.whbdense_stride;
	const unsigned long long *xbits = T.whbxbits + (size_t)s * T.whbx_stride;
	const double2 *xsnap = T.whbxsnap + (size_t)s * T.whbx_stride;
	const WhbExact st0 = T.whbx[s];
	const int tp_ = whb_hook_perturb(T);
	const int tol = tp_ > 1 ? tp_ : (tp_ < -1 ? -tp_ : 1);
	auto input = [&](int pos) -> int { return pos >= 0 ? dense[pos] : (pos == -1 ? st0.fd1 : st0.fd2); };
	// the chain over `cnt` inputs from the first of round rr, by the calling lane: the state behind them and their decisions
	auto run = [&](int rr, int cnt, double &y1, double &y2, unsigned long long &bits) {
		const int p0 = 64 * rr;
		if (rr == 0) {
			y1 = st0.y1;
			y2 = st0.y2;
		} else {
			const double2 sn = xsnap[rr - 1];
			y1 = sn.x;
			y2 = sn.y;
		}
		double t1 = bh * (double)input(p0 - 1), t2 = bh * (double)input(p0 - 2);
		bits = 0ull;
		for (int k = 0; k < cnt; k++) {
			const int d = dense[p0 + k];
			const double t0 = bh * (double)d;
			const double P = __builtin_fma(2.0, t1, t0);
			const double y = ((t2 + a1 * y1) + P) + a2 * y2;
			bits |= (unsigned long long)(d < (int)y ? 1u : 0u) << k;
			y2 = y1;
			y1 = y;
			t2 = t1;
			t1 = t0;
		}
	};
	// ---- the inputs behind the last whole round: the filter's state at the end of the submit
	double yf1 = 0.0, yf2 = 0.0;
	unsigned long long tailbits = 0ull;
	if (ln == 0)
		run(R, N & 63, yf1, yf2, tailbits);
	tailbits = __shfl(tailbits, 0, 64);
	auto word = [&](int i) -> unsigned long long { return i < R ? xbits[i] : (i == R ? tailbits : 0ull); };
	// ---- the records, 64 at a time, a lane each
	const uint4 *recrow = reinterpret_cast<const uint4 *>(T.whbrec + (size_t)s * T.whbrec_stride);
	int carry = carry_io[s];  // exact minus speculated frozen average of a window still open and locked (0, +1, -1)
	const int carry_in = carry;
	bool bad = false;
	int cbase = 0, nlocks = 0;
	bool ended = false;
	int last_delta = 0;
	bool have_last = false, last_closed = false;
	auto do_locks = [&]() {  // the collected locks, a lane each: (int) of the exact average against the frozen integer
		for (int b0 = 0; b0 < nlocks; b0 += 64) {
			const int i = b0 + ln;
			const bool on = i < nlocks;
			double y1 = 0.0, y2 = 0.0;
			unsigned long long bits;
			const int pos = on ? lk_pos[i] : 0;
			run(pos >> 6, on ? (pos & 63) + 1 : 0, y1, y2, bits);
			const uint32_t meta = on ? lk_meta[i] : 0u;
			const int delta = (int)y1 - (on ? lk_avgf[i] : 0);
			const bool b = on && (delta > tol || delta < -tol || (delta != 0 && (meta & kWhbRecAmb)));
			bad = bad || __ballot(b) != 0ull;
			// the carry is the last lock's (in record order)
			const int nb = nlocks - b0 < 64 ? nlocks - b0 : 64;
			last_delta = __shfl(delta, nb - 1, 64);
			last_closed = (__shfl((int)meta, nb - 1, 64) & (int)kWhbRecClosed) != 0;
			have_last = true;
		}
		nlocks = 0;
	};
	for (int v0 = 0; !ended; v0 += 64) {
		const int vi = v0 + ln < T.whbrec_stride - 1 ? v0 + ln : T.whbrec_stride - 1;  // (lanes behind the end mark read anything inside the row)
		const uint4 rec = recrow[vi];
		const uint32_t meta = rec.z;
		const unsigned long long endm = __ballot(meta == kWhbRecEnd);
		const int nrec = endm ? __builtin_ctzll(endm) : 64;
		ended = endm != 0ull || v0 + 64 + 64 > T.whbrec_stride;
		const bool on = ln < nrec;
		const bool pseudo = on && (meta & kWhbRecPseudo) != 0u;
		const bool normal = on && !pseudo;
		const int nv = normal ? (int)((meta >> kWhbRecNvShift) & 63u) + 1 : 0;
		// exclusive prefix sum of the lengths: the record's first input in the sequence
		int inc = nv;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int up = __shfl_up(inc, o, 64);
			inc += ln >= o ? up : 0;
		}
		const int pos0 = cbase + inc - nv;
		cbase += __shfl(inc, 63, 64);
		if (pseudo) {  // a window that began locked (it continues one of the submit before): no filter step -- only ever the first record
			if (v0 + ln == 0) {
				bad = bad || (carry != 0 && (meta & kWhbRecAmb));
				if (meta & kWhbRecClosed)
					carry = 0;
			} else {
				bad = true;
			}
		}
		if (normal) {
			const int w = pos0 >> 6, sh = pos0 & 63;
			const unsigned long long lo = word(w) >> sh, hi = sh ? word(w + 1) << (64 - sh) : 0ull;
			const unsigned long long vm = nv >= 64 ? ~0ull : (1ull << nv) - 1ull;
			const unsigned long long below = ((unsigned long long)rec.y << 32) | rec.x;
			bad = bad || (((lo | hi) ^ below) & vm) != 0ull || pos0 + nv > N;
		}
		// (what lane 0 decided about a pseudo record: one value for the wave)
		carry = __shfl(carry, 0, 64);
		// the chunk's locks join the list
		const bool lock = normal && (meta & kWhbRecLock) != 0u;
		const unsigned long long lm = __ballot(lock);
		if (lm) {
			if (nlocks + 64 > kChkLockCap)
				do_locks();
			if (lock) {
				const int i = nlocks + __builtin_popcountll(lm & ((1ull << ln) - 1ull));
				lk_pos[i] = pos0 + nv - 1;
				lk_avgf[i] = (int)rec.w;
				lk_meta[i] = meta;
			}
			nlocks += __builtin_popcountll(lm);
			__syncthreads();
		}
	}
	if (nlocks)
		do_locks();
	bad = __ballot(bad) != 0ull || cbase != N;
	if (have_last)
		carry = last_closed ? 0 : last_delta;
	if (ln == 0) {
		WhbExact st = st0;
		st.carry = carry_in;
		st.pad_ = 0;
		T.whbx0[s] = st;  // the exact state this submit started from, and the carry (a redo needs both)
		// a stream whose speculative pass started from a state that a redo has replaced since is redone as well
		bad = bad || T.whbseen[s] != T.whbgen[s];
		if (whb_hook_force_fail(T) > 0 && (s + T.whb_submit_seq) % whb_hook_force_fail(T) == 0)
			bad = true;  // tests
		st.y1 = yf1;
		st.y2 = yf2;
		st.fd1 = input(N - 1);
		st.fd2 = input(N - 2);
		st.carry = st.pad_ = 0;
		T.whbx[s] = st;
		carry_io[s] = bad ? 0 : carry;  // (the exact kernel freezes the exact average: nothing to carry)
		T.whbfail[s] = bad ? 1 : 0;
	}
}

