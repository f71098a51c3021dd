// tfrec_amd/csrc/whb_commit.h -- K4'' whb_decoder::store_bit over the accepted runs and the stream's flush events (called from whb_demod_kernel's tail).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

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
