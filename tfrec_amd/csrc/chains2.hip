// tfrec_amd/csrc/chains2.hip -- window-parallel demodulator/decoder pipeline (the product path).
//
// Same results as the serial reference chains of chains.hip (and therefore as fsk_demod::process +
// the plugins, fm_demod.cpp:34-56, tfa1.cpp:143-190, tfa2.cpp:346-442, whb.cpp:632-707), re-cut so that
// only what the reference really serialises stays serial:
//
//   K2 windows_kernel   lane per (stream, slot): max-scan over the trigger mask -> list of trigger windows
//                       (tfa1.cpp:147-149,179 / tfa2.cpp:351-355,428 / whb.cpp:636-641,691: a window opens at a
//                       pwr>thresh sample while the counter is 0 and its flush fires W-1 samples after the
//                       last trigger).
//   K3 biquad_kernel    lane per (stream, slot): the fp64 biquads (iir2::step) are the one recurrence whose
//                       state crosses windows; this pass does nothing else: 8 samples per 16-byte load,
//                       TFA_2/TFA_3/TX22 read the shared fm_dev array of the front-end, WHB computes
//                       fm_dev_nrzs on the fly; outputs are the truncated integers the slicers consume.
//   K4 slicer_kernel    lane per WINDOW (work queue, long windows first): the bit slicers of TFA_1 and the
//                       TFA_2 family are window-local state machines (state reset at window open/close) --
//                       except tfa2's last_bit_idx, which is never reset (tfa2.cpp:325-334).  It can only
//                       influence a window through its first candidate edge, so windows are run assuming a
//                       far-away last edge and the assumption is checked (and the window re-run exactly) in K5.
//                       Output: the bits handed to decoder::store_bit, packed.
//   K4' whb_kernel      lane per stream: WHB stage 2 (decision-level biquad, phase-change detector, decoder);
//                       demodulator and decoder feed back into each other through has_sync() (whb.cpp:653,
//                       677, 693), so this chain stays serial.
//   K5 commit_kernel    lane per (stream, slot): walks the windows in order: validates/repairs the tfa2
//                       speculation, runs the decoders (store_bit / flush) over the packed bits with their
//                       persistent state (sr, rdata), emits events, commits ChainState for the next submit.
#include "decoder_dev.h"

namespace tfrec {

constexpr int kSpecLbi = -(1 << 30);  // "last edge far in the past"

// demodulator::start (decoder.cpp:118-122) applied once per block boundary between two blocks
__device__ __forceinline__ int rebase_lbi(int lbi, int from_block, int to_block)
{
	return lbi ? lbi - kIndexSpan * (to_block - from_block) : 0;
}

struct BitWriter {
	uint32_t *base;
	uint32_t acc;
	int n;
	__device__ __forceinline__ void put(int bit)
	{
		acc |= (uint32_t)bit << (n & 31);
		n++;
		if ((n & 31) == 0) {
			base[(n >> 5) - 1] = acc;
			acc = 0;
		}
	}
	__device__ __forceinline__ void finish()
	{
		if (n & 31)
			base[n >> 5] = acc;
	}
};

// ------------------------------------------------------------------------------------------------ K2
__global__ __launch_bounds__(64) void windows_kernel(const unsigned long long *__restrict__ mask, size_t mask_stride,
						     int n_streams, int n_blocks, ChainLaunch L, WinTables T)
{
	const int a = blockIdx.y;
	const int s = blockIdx.x * 64 + threadIdx.x;
	if (s >= n_streams)
		return;
	const ChainParams &p = L.params[a];
	const int c = a * n_streams + s;
	const int W = p.window;
	const int M = n_blocks * kBlockDec;
	const int nwords = M >> 6;
	const unsigned long long *mrow = mask + (size_t)s * mask_stride;
	const int t0 = L.states[a][s].timeout_cnt;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;

	bool open = t0 > 0;
	int open_g = 0;
	int last_trig = open ? t0 - W : -(1 << 29);  // virtual trigger that leaves t0 samples of window
	int count = 0;
	bool overflow = false;
	auto emit = [&](int og, int close) {
		if (count < T.cap) {
			T.open[(size_t)c * T.cap + count] = og;
			T.close[(size_t)c * T.cap + count] = close;
			if (p.kind < 2) {
				const int last = close < M ? close : M - 1;
				const int q = 2 * p.kind + ((last - og + 1) >= kLongWindow ? 0 : 1);
				const uint32_t idx = atomicAdd(&T.queue[q].count, 1u);
				T.items[(size_t)q * total + idx] = make_uint2((uint32_t)c, (uint32_t)count);
			}
			count++;
		} else
			overflow = true;
	};
	for (int w = 0; w < nwords; w++) {
		const unsigned long long m = mrow[w];
		if (!m)
			continue;
		// a gap that closes a window is >= W-1 >= 355 samples, so it always spans whole words:
		// only the first and last trigger of a non-zero word matter
		const int first = (w << 6) + __builtin_ctzll(m);
		if (open && first > last_trig + W - 1) {
			emit(open_g, last_trig + W - 1);
			open = false;
		}
		if (!open) {
			open = true;
			open_g = first;
		}
		last_trig = (w << 6) + 63 - __builtin_clzll(m);
	}
	int tnext = 0;
	if (open) {
		const int close = last_trig + W - 1;
		emit(open_g, close);
		if (close >= M)
			tnext = close - (M - 1);
	}
	T.count[c] = count;
	T.cont[c] = t0 > 0 ? 1 : 0;
	T.timeout_next[c] = tnext;
	if (overflow)
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
// One lane = one biquad chain; double-buffered 32-sample chunks in registers (the loads of chunk n+1 are in
// flight while chunk n is filtered), fully unrolled predicated steps.
__device__ __forceinline__ void k3_load16(const int16_t *row, const ChunkDesc &d, uint4 (&b)[4])
{
	const uint4 *p = reinterpret_cast<const uint4 *>(row + d.cb);
#pragma unroll
	for (int i = 0; i < 4; i++)
		b[i] = p[i];
}

__device__ __forceinline__ void k3_proc16(Biquad &f, const BiquadCoef &cf, const ChunkDesc &d, const uint4 (&b)[4],
					  int16_t *out)
{
	const bool full = (d.lo == d.cb) && (d.hi == d.cb + kChunk - 1);
	uint32_t ow[16];
#pragma unroll
	for (int i = 0; i < 16; i++)
		ow[i] = 0;
#pragma unroll
	for (int k = 0; k < kChunk; k++) {
		const int gk = d.cb + k;
		const uint4 &q = b[k >> 3];
		const uint32_t w = ((k >> 1) & 3) == 0 ? q.x : ((k >> 1) & 3) == 1 ? q.y : ((k >> 1) & 3) == 2 ? q.z : q.w;
		if (gk >= d.lo && gk <= d.hi) {
			const int x = (int)(int16_t)((w >> (16 * (k & 1))) & 0xffff);
			const int y = d2i(iir_step(f, cf, (double)x));
			ow[k >> 1] |= ((uint32_t)y & 0xffffu) << (16 * (k & 1));
			if (!full)
				out[gk] = (int16_t)y;
		}
	}
	if (full) {
		uint4 *o = reinterpret_cast<uint4 *>(out + d.cb);
#pragma unroll
		for (int i = 0; i < 4; i++)
			o[i] = make_uint4(ow[4 * i], ow[4 * i + 1], ow[4 * i + 2], ow[4 * i + 3]);
	}
}

__device__ __forceinline__ void k3_load32(const uint32_t *drow, const ChunkDesc &d, uint4 (&b)[8], uint32_t &prevw,
					  uint32_t prev0)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(drow + d.cb);
#pragma unroll
	for (int i = 0; i < 8; i++)
		b[i] = p[i];
	prevw = d.cb > 0 ? drow[d.cb - 1] : prev0;
}

__device__ __forceinline__ void k3_proc32(Biquad &f, const BiquadCoef &cf, const ChunkDesc &d, const uint4 (&b)[8],
					  uint32_t prevw, int32_t *out)
{
	const bool full = (d.lo == d.cb) && (d.hi == d.cb + kChunk - 1);
	int pI = (int)(int16_t)(prevw & 0xffff), pQ = (int)prevw >> 16;
	int ow[kChunk];
#pragma unroll
	for (int k = 0; k < kChunk; k++) {
		const int gk = d.cb + k;
		const uint4 &q = b[k >> 2];
		const uint32_t w = (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
		const int I = (int)(int16_t)(w & 0xffff), Q = (int)w >> 16;
		ow[k] = 0;
		if (gk >= d.lo && gk <= d.hi) {
			// WHB stage 1: dev = (int) iir->step(fm_dev_nrzs(iq, last_iq)), whb.cpp:651-652
			ow[k] = d2i(iir_step(f, cf, (double)fm_dev_nrzs(I, Q, pI, pQ)));
			if (!full)
				out[gk] = ow[k];
		}
		pI = I;
		pQ = Q;
	}
	if (full) {
		int4 *o = reinterpret_cast<int4 *>(out + d.cb);
#pragma unroll
		for (int i = 0; i < 8; i++)
			o[i] = make_int4(ow[4 * i], ow[4 * i + 1], ow[4 * i + 2], ow[4 * i + 3]);
	}
}

__global__ __launch_bounds__(64) void biquad_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						    const int16_t *__restrict__ fmdev, size_t fmdev_stride, int n_streams,
						    int n_blocks, ChainLaunch L, WinTables T, int16_t *__restrict__ ld16,
						    int32_t *__restrict__ dev32)
{
	const int a = blockIdx.y;
	const int s = blockIdx.x * 64 + threadIdx.x;
	const ChainParams &p = L.params[a];
	if (s >= n_streams || p.kind == 0)
		return;
	const int c = a * n_streams + s;
	const int M = n_blocks * kBlockDec;
	ChainState &st = L.states[a][s];
	Biquad f = st.iir;
	const BiquadCoef cf = p.iir;
	ChunkIter it;
	it.init(T, c, M);
	ChunkDesc dA, dB;
	if (p.kind == 1) {
		const int16_t *in = fmdev + (size_t)s * fmdev_stride;
		int16_t *out = ld16 + (size_t)c * M;
		uint4 A[4], B[4];
		bool hA = it.next(dA), hB;
		if (hA)
			k3_load16(in, dA, A);
		while (hA) {
			hB = it.next(dB);
			if (hB)
				k3_load16(in, dB, B);
			k3_proc16(f, cf, dA, A, out);
			if (!hB)
				break;
			hA = it.next(dA);
			if (hA)
				k3_load16(in, dA, A);
			k3_proc16(f, cf, dB, B, out);
		}
	} else {
		const uint32_t *drow = dec + (size_t)s * dec_stride;
		int32_t *out = dev32 + (size_t)s * M;
		const uint32_t prev0 = ((uint32_t)st.prev_i & 0xffffu) | ((uint32_t)st.prev_q << 16);
		uint4 A[8], B[8];
		uint32_t pA = 0, pB = 0;
		bool hA = it.next(dA), hB;
		if (hA)
			k3_load32(drow, dA, A, pA, prev0);
		while (hA) {
			hB = it.next(dB);
			if (hB)
				k3_load32(drow, dB, B, pB, prev0);
			k3_proc32(f, cf, dA, A, pA, out);
			if (!hB)
				break;
			hA = it.next(dA);
			if (hA)
				k3_load32(drow, dA, A, pA, prev0);
			k3_proc32(f, cf, dB, B, pB, out);
		}
	}
	st.iir = f;
}

// ------------------------------------------------------------------------------------------------ slicers

struct Slicer {  // window-local demodulator state (tfa1.h:28-32, tfa2.h:35-42)
	int lbi;     // last_bit_idx, relative to block cur_block
	int cur_block;
	int mark_lvl, rssi_i;                          // tfa1 (rssi_i also tfa2)
	int bitcnt, dmin, dmax, offset, last_bit;      // tfa2
	int first_cand_g;
};

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
	(void)kind;
}

// One sample of tfa1_demod::demod inside a window (tfa1.cpp:150-178); the flush at the window's last sample
// is done by the caller.  (BITPERIOD 10: ones are emitted for n = 22, 42, ... <= gap.)
__device__ __forceinline__ void tfa1_sample(Slicer &f, BitWriter &bw, int g, int I, int Q, int pI, int pQ)
{
	const int b = g >> 13;
	if (b != f.cur_block) {
		f.lbi = rebase_lbi(f.lbi, f.cur_block, b);
		f.cur_block = b;
	}
	const int index = 2 * (g & (kBlockDec - 1));
	const int dev = fm_dev_nrzs(I, Q, pI, pQ);
	if (dev > f.mark_lvl)
		f.mark_lvl = dev;
	else
		f.mark_lvl = d2i(f.mark_lvl * 0.95);
	if (f.mark_lvl > f.rssi_i)
		f.rssi_i = f.mark_lvl;
	if (dev < f.mark_lvl / 2) {
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

// One sample of tfa2_demod::demod inside a window (tfa2.cpp:357-412), ld = (int)iir->step(fm_dev(...)).
__device__ __forceinline__ void tfa2_sample(Slicer &f, BitWriter &bw, int g, int ld, const uint32_t *drow, double spb)
{
	const int b = g >> 13;
	if (b != f.cur_block) {
		f.lbi = rebase_lbi(f.lbi, f.cur_block, b);
		f.cur_block = b;
	}
	const int index = 2 * (g & (kBlockDec - 1));
	if (f.bitcnt < 10) {
		if (ld > f.dmax)
			f.dmax = (7 * f.dmax + ld) / 8;
		if (ld < f.dmin)
			f.dmin = (7 * f.dmin + ld) / 8;
		f.offset = (f.dmax + f.dmin) / 2;
		if (f.bitcnt > 4) {  // wrapping int32 arithmetic as in the reference binary (tfa2.cpp:373)
			const uint32_t cw = drow[g];
			const int I = (int)(int16_t)(cw & 0xffff), Q = (int)cw >> 16;
			const uint32_t t = (uint32_t)f.rssi_i + (uint32_t)(I * I) + (uint32_t)(Q * Q);
			f.rssi_i = (int)((uint32_t)f.rssi_i + (uint32_t)((int)t / 100));
		}
	}
	const int noffset = d2i(0.9 * f.offset);
	const int hi = noffset + f.dmax / 32, lo = noffset + f.dmin / 32;
	const int bit = ld > hi ? 1 : 0;
	if ((ld > hi || ld < lo) && bit != f.last_bit) {
		if (f.first_cand_g < 0)
			f.first_cand_g = g;
		if (index > f.lbi + 8) {
			f.bitcnt++;
			const int tdiff = index - f.lbi;
			if (tdiff > spb / 4 && tdiff < 32 * spb) {
				const int numbits = d2i(((tdiff / 2) + (spb / 2)) / spb);
				if (numbits < 32)
					for (int n = 1; n < numbits; n++)
						bw.put(f.last_bit);
				bw.put(bit);
				f.last_bit = bit;
			}
		}
		if (index - f.lbi > 2)
			f.lbi = index;
	}
}

// Run one window [g0, last] of a TFA_1 (KIND 0) or TFA_2-family (KIND 1) slicer.  `f` carries the state in and
// out; bits go to bw.  Returns with f.cur_block = block of `last`.
template <int KIND>
__device__ __forceinline__ void run_window(Slicer &f, BitWriter &bw, int g0, int last, bool closed,
					   const uint32_t *__restrict__ drow, const int16_t *__restrict__ ldrow, int prevI,
					   int prevQ, double spb)
{
	int g = g0;
	while (g <= last) {
		const int cb = g & ~7;
		if (KIND == 0) {
			const uint4 v0 = *reinterpret_cast<const uint4 *>(drow + cb);
			const uint4 v1 = *reinterpret_cast<const uint4 *>(drow + cb + 4);
			const uint32_t vw[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
			int pI = prevI, pQ = prevQ;
			if (cb > 0) {
				const uint32_t pw = drow[cb - 1];
				pI = (int)(int16_t)(pw & 0xffff);
				pQ = (int)pw >> 16;
			}
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int gk = cb + k;
				const int I = (int)(int16_t)(vw[k] & 0xffff), Q = (int)vw[k] >> 16;
				if (gk >= g && gk <= last)
					tfa1_sample(f, bw, gk, I, Q, pI, pQ);
				pI = I;
				pQ = Q;
			}
		} else {
			const uint4 v = *reinterpret_cast<const uint4 *>(ldrow + cb);
			const uint32_t vw[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const int gk = cb + k;
				if (gk >= g && gk <= last) {
					const int ld = (int)(int16_t)((vw[k >> 1] >> (16 * (k & 1))) & 0xffff);
					tfa2_sample(f, bw, gk, ld, drow, spb);
				}
			}
		}
		g = cb + 8;
	}
	const int bl = last >> 13;
	if (bl != f.cur_block) {  // no sample processed in the last block?  cannot happen (last is processed), kept for safety
		f.lbi = rebase_lbi(f.lbi, f.cur_block, bl);
		f.cur_block = bl;
	}
	if (closed && KIND == 1)  // tfa2.cpp:430-431: trailing bits before the flush
		for (int n = 0; n < 16; n++)
			bw.put(f.last_bit);
}

template <int KIND>
__device__ __forceinline__ void window_task(int c, int j, int n_streams, int M, const uint32_t *__restrict__ dec,
					    size_t dec_stride, const int16_t *__restrict__ ld16, const ChainLaunch &L,
					    const WinTables &T, bool exact_lbi, int lbi_in_override)
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
	BitWriter bw{ T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j, 0u, 0 };
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const int16_t *ldrow = (KIND == 1) ? ld16 + (size_t)c * M : nullptr;
	run_window<KIND>(f, bw, og, last, closed, drow, ldrow, st.prev_i, st.prev_q, p.spb);
	bw.finish();
	WinResult &r = T.result[(size_t)c * T.cap + j];
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
// Persistent lanes pull (chain, window) items, long windows first.  blockIdx.y = protocol kind (0 TFA_1,
// 1 TFA_2 family), each with its own pair of queues, so a wave runs one slicer type.
__global__ __launch_bounds__(64) void slicer_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						    const int16_t *__restrict__ ld16, int n_streams, int n_blocks, ChainLaunch L,
						    WinTables T)
{
	const int M = n_blocks * kBlockDec;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	const int kind = blockIdx.y;
	for (int q = 2 * kind; q < 2 * kind + 2; q++) {
		const uint32_t count = T.queue[q].count;
		while (true) {
			const uint32_t idx = atomicAdd(&T.queue[q].head, 1u);
			if (idx >= count)
				break;
			const uint2 it = T.items[(size_t)q * total + idx];
			const int c = (int)it.x, j = (int)it.y;
			if (kind == 0)
				window_task<0>(c, j, n_streams, M, dec, dec_stride, ld16, L, T, false, 0);
			else
				window_task<1>(c, j, n_streams, M, dec, dec_stride, ld16, L, T, false, 0);
		}
	}
}

// ------------------------------------------------------------------------------------------------ K4' WHB stage 2
// Serial lane per stream.  Per iteration: the current 32-sample chunk (stage-1 output + decimated IQ) is moved
// from registers to a lane-private LDS column, the NEXT chunk's global loads are issued, then the chunk is
// walked sample by sample from LDS -- so HBM latency overlaps the state machine instead of preceding it.
__global__ __launch_bounds__(64) void whb_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						 const int32_t *__restrict__ dev32, int n_streams, int n_blocks, long long sample_base,
						 ChainLaunch L, int a, WinTables T, tfrec_amd_event *__restrict__ events,
						 EventBuf *__restrict__ eb, uint32_t flags)
{
	__shared__ uint32_t sdev[kChunk * 64];
	__shared__ uint32_t sdec[kChunk * 64];
	const int lane = threadIdx.x;
	const int s = blockIdx.x * 64 + lane;
	if (s >= n_streams)
		return;
	const ChainParams &p = L.params[a];
	ChainState &st = L.states[a][s];
	const int c = a * n_streams + s;
	const int M = n_blocks * kBlockDec;
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const int32_t *dvrow = dev32 + (size_t)s * M;
	EmitCtx e{ events, eb, flags, (uint32_t)s, L.slot[a], p.sensor_type, sample_base };
	Dec d{ st.sr, st.sr_cnt, st.byte_cnt, st.invert, st.synced, st.w_last_bit, st.psk, st.last_psk, st.nrzs, st.lfsr, st.seq,
	       st.rdata };
	int last_dev = st.last_dev, avg_of = st.avg_of;
	unsigned long long step = st.step, last_peak = st.last_peak;
	double rssi_d = st.rssi_d;
	Biquad iir_avg = st.iir_avg;
	const BiquadCoef cavg = p.iir_avg;
	const double spb = p.spb;
	const bool cont = T.cont[c] != 0;

	ChunkIter it;
	it.init(T, c, M);
	ChunkDesc dC, dN;
	uint4 rv[8], rd[8];
	auto load = [&](const ChunkDesc &dd) {
		const uint4 *pv = reinterpret_cast<const uint4 *>(dvrow + dd.cb);
		const uint4 *pd = reinterpret_cast<const uint4 *>(drow + dd.cb);
#pragma unroll
		for (int i = 0; i < 8; i++) {
			rv[i] = pv[i];
			rd[i] = pd[i];
		}
	};
	bool hC = it.next(dC);
	if (hC)
		load(dC);
	while (hC) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			sdev[(4 * i + 0) * 64 + lane] = rv[i].x;
			sdev[(4 * i + 1) * 64 + lane] = rv[i].y;
			sdev[(4 * i + 2) * 64 + lane] = rv[i].z;
			sdev[(4 * i + 3) * 64 + lane] = rv[i].w;
			sdec[(4 * i + 0) * 64 + lane] = rd[i].x;
			sdec[(4 * i + 1) * 64 + lane] = rd[i].y;
			sdec[(4 * i + 2) * 64 + lane] = rd[i].z;
			sdec[(4 * i + 3) * 64 + lane] = rd[i].w;
		}
		const bool hN = it.next(dN);
		if (hN)
			load(dN);
		if ((dC.flags & 1) && !(dC.j == 0 && cont)) {  // window opens: whb_demod::reset, whb.cpp:616-623
			rssi_d = 0;
			step = last_peak = 0;
		}
		const int k1 = dC.hi - dC.cb;
		for (int k = dC.lo - dC.cb; k <= k1; k++) {
			const int dev = (int)sdev[k * 64 + lane];  // stage 1 (biquad_kernel)
			if (!d.synced)
				avg_of = d2i(iir_step(iir_avg, cavg, 0.5 * dev));
			const int tdiff = (int)(step - last_peak);
			if (dev < avg_of && dev > last_dev && (tdiff > 3 * spb / 4)) {  // phase change, whb.cpp:662-673
				store_bit<2>(d, 0);
				const int bit0 = d2i((tdiff + spb / 2) / spb);
				for (int n = 1; n < bit0; n++)
					store_bit<2>(d, 1);
				last_peak = step;
			}
			last_dev = dev;
			if (d.synced) {
				const uint32_t cw = sdec[k * 64 + lane];
				const int I = (int)(int16_t)(cw & 0xffff), Q = (int)cw >> 16;
				rssi_d += (double)(I * I + Q * Q);
			}
			if (k == k1 && (dC.flags & 4)) {  // timeout_cnt reached 0, whb.cpp:691-702
				if (d.synced) {
					for (int n = 0; n < 16; n++)
						store_bit<2>(d, 0);
					flush<2>(e, d, (long long)rssi_d, 0, dC.cb + k);
				}
				rssi_d = 0;
				step = last_peak = 0;
			}
			step++;
		}
		dC = dN;
		hC = hN;
	}
	{
		const uint32_t lw = drow[M - 1];
		st.prev_i = (int)(int16_t)(lw & 0xffff);
		st.prev_q = (int)lw >> 16;
	}
	st.timeout_cnt = T.timeout_next[c];
	st.last_dev = last_dev;
	st.avg_of = avg_of;
	st.step = step;
	st.last_peak = last_peak;
	st.rssi_d = rssi_d;
	st.iir_avg = iir_avg;
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

// ------------------------------------------------------------------------------------------------ K5
template <int KIND>
__device__ __forceinline__ void commit_body(int a, int s, int n_streams, int n_blocks, long long sample_base,
					    const uint32_t *__restrict__ dec, size_t dec_stride,
					    const int16_t *__restrict__ ld16, const ChainLaunch &L, const WinTables &T,
					    tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags)
{
	const int M = n_blocks * kBlockDec;
	const ChainParams &p = L.params[a];
	ChainState &st = L.states[a][s];
	const int c = a * n_streams + s;
	const int count = T.count[c];
	EmitCtx e{ events, eb, flags, (uint32_t)s, L.slot[a], p.sensor_type, sample_base };
	Dec d{ st.sr, st.sr_cnt, st.byte_cnt, st.invert, st.synced, st.w_last_bit, st.psk, st.last_psk, st.nrzs, st.lfsr, st.seq,
	       st.rdata };
	int lbi = st.last_bit_idx;  // true last_bit_idx, relative to lbi_block
	int lbi_block = -1;
	const WinResult *last_r = nullptr;
	for (int j = 0; j < count; j++) {
		const int og = T.open[(size_t)c * T.cap + j];
		const int close = T.close[(size_t)c * T.cap + j];
		const int last = close < M ? close : M - 1;
		WinResult *r = &T.result[(size_t)c * T.cap + j];
		if (KIND == 1) {
			if (j > 0) {
				// window j was sliced assuming last_bit_idx far in the past (kSpecLbi); check with the true value
				if (r->first_cand_g >= 0) {
					const int bc = r->first_cand_g >> 13;
					const int index_c = 2 * (r->first_cand_g & (kBlockDec - 1));
					const int lbi_c = rebase_lbi(lbi, lbi_block, bc);
					const int tdiff = index_c - lbi_c;
					// the speculative run saw: glitch test passed, edge counted, nothing emitted, last_bit kept
					// (tfa2.cpp:391-409 with a huge tdiff).  The true run does the same iff:
					const bool same = (index_c > lbi_c + 8) && !(tdiff > p.spb / 4 && tdiff < 32 * p.spb);
					if (!same)
						window_task<1>(c, j, n_streams, M, dec, dec_stride, ld16, L, T, true,
							       rebase_lbi(lbi, lbi_block, og >> 13));
					lbi = r->lbi_out;
				} else {
					lbi = rebase_lbi(lbi, lbi_block, last >> 13);  // no candidate edge: it just ages
				}
			} else {
				lbi = r->lbi_out;  // window 0 always runs with the exact carried value
			}
			lbi_block = last >> 13;
		}
		// decoder over the window's bits (decoder::store_bit), then decoder::flush if the window closed
		const uint32_t *bits = T.bits + (size_t)c * T.bit_words + (og >> 6) + 3 * j;
		const int nbits = r->nbits;
		uint32_t wbits = 0;
		for (int n = 0; n < nbits; n++) {
			if ((n & 31) == 0)
				wbits = bits[n >> 5];
			store_bit<KIND>(d, (wbits >> (n & 31)) & 1);
		}
		if (r->closed)
			flush<KIND>(e, d, r->rssi_i, KIND == 1 ? r->offset : 0, last);
		last_r = r;
	}
	// ---- commit the state the next submit starts from
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
						    tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags)
{
	const int a = blockIdx.y;
	const int s = blockIdx.x * 64 + threadIdx.x;
	if (s >= n_streams)
		return;
	const int kind = L.params[a].kind;
	if (kind == 0)
		commit_body<0>(a, s, n_streams, n_blocks, sample_base, dec, dec_stride, ld16, L, T, events, eb, flags);
	else if (kind == 1)
		commit_body<1>(a, s, n_streams, n_blocks, sample_base, dec, dec_stride, ld16, L, T, events, eb, flags);
}

// ------------------------------------------------------------------------------------------------ launch
hipError_t launch_pipeline(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			   size_t mask_stride, const int16_t *fmdev, size_t fmdev_stride, int n_streams, int n_blocks,
			   long long sample_base, const ChainLaunch &L, const WinTables &T, int16_t *ld16, int32_t *dev32,
			   tfrec_amd_event *events, EventBuf *eb, uint32_t flags, int slicer_waves)
{
	if (L.n_active == 0)
		return hipSuccess;
	hipError_t e = hipMemsetAsync(T.queue, 0, 4 * sizeof(WorkQueue), st);
	if (e != hipSuccess)
		return e;
	dim3 grid((n_streams + 63) / 64, L.n_active), block(64);
	hipLaunchKernelGGL(windows_kernel, grid, block, 0, st, mask, mask_stride, n_streams, n_blocks, L, T);
	hipLaunchKernelGGL(biquad_kernel, grid, block, 0, st, dec, dec_stride, fmdev, fmdev_stride, n_streams, n_blocks, L, T,
			   ld16, dev32);
	hipLaunchKernelGGL(slicer_kernel, dim3(slicer_waves, 2), block, 0, st, dec, dec_stride, ld16, n_streams, n_blocks, L, T);
	for (int a = 0; a < L.n_active; a++)
		if (L.params[a].kind == 2)
			hipLaunchKernelGGL(whb_kernel, dim3((n_streams + 63) / 64), block, 0, st, dec, dec_stride, dev32, n_streams,
					   n_blocks, sample_base, L, a, T, events, eb, flags);
	hipLaunchKernelGGL(commit_kernel, grid, block, 0, st, dec, dec_stride, ld16, n_streams, n_blocks, sample_base, L, T,
			   events, eb, flags);
	return hipGetLastError();
}

}  // namespace tfrec
