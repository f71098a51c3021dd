// tfrec_amd/csrc/decode.h -- K5 decode_kernel / commit_kernel / commit_wave_kernel: decoder::store_bit and flush of TFA_1 and the TFA_2 family.
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
