// tfrec_amd/csrc/coop_slicer.h -- K4b coop_slicer_kernel: wave per LONG window (scalar walks and the step-per-lane forms).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
