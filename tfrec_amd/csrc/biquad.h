// tfrec_amd/csrc/biquad.h -- the chunk iterator of the serial lanes and K3: spec_biquad_kernel / fix_biquad_kernel (iir2::step chains, speculate + repair).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
