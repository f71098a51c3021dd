// tfrec_amd/csrc/slicers.h -- the slicer rules of TFA_1 and the TFA_2 family, K4 slicer_kernel (lane per window) and K4a' mark_kernel.
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
