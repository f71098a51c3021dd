// tfrec_amd/csrc/chains.hip -- demodulator + decoder chains: decimated IQ + trigger mask -> flush events.
//
// Replaces the reference's per-sample loop of fsk_demod::process (fm_demod.cpp:34-56) together with the
// plugin bodies it dispatches to:
//   demodulator::start            decoder.cpp:118-122  (last_bit_idx rebase per 8192-sample block)
//   tfa1_demod::demod             tfa1.cpp:143-190     tfa1_decoder::store_bit tfa1.cpp:120-134
//   tfa2_demod::demod / ::reset   tfa2.cpp:325-442     tfa2_decoder::store_bit tfa2.cpp:281-314
//   whb_demod::demod / ::reset    whb.cpp:616-707      whb_decoder::store_bit  whb.cpp:566-603
//   fm_dev_nrzs / fm_dev          dsp_stuff.cpp:269-292, iir2::step dsp_stuff.cpp:47-56
// and evaluates the acceptance tests of the decoders' flush() (CRC-8 / CRC-32 + sanity:
// tfa1.cpp:66-73, tfa2.cpp:84-93, 237, whb.cpp:484-510) so that every event carries a verdict.
//
// Mapping (round 1): one lane per (stream, demodulator slot); a wave holds 64 streams of the same slot so
// the control flow of a wave is one protocol.  Each lane walks ITS OWN stream position: samples outside a
// trigger window are skipped 64 at a time through the trigger mask (ctz over 64-bit words), samples inside
// a window run the reference's state machine verbatim.  All recurrences that the reference carries across
// windows and blocks (biquad state, last_bit_idx with its block rebase and ==0 sentinel, decoder shift
// registers, persistent rdata) are carried in ChainState, so results are independent of how the stream is
// cut into submits.
//
// Floating point: strict IEEE fp64, no contraction (built with -ffp-contract=off), in the evaluation
// order of the reference's normative -ffast-math x86-64 build (see oracle/tfrec_oracle.c header):
// the biquad is ((b2*dn2 + a1*yn1) + (b0*dn + b1*dn1)) + a2*yn2.
#include "decoder_dev.h"

namespace tfrec {

__device__ __constant__ double kAtanPolySerial[16] = TFREC_ATAN_POLY;  // see dsp_dev.h

template <int KIND>
__device__ __forceinline__ void chain_body(const uint32_t *__restrict__ dec, size_t dec_stride,
					   const unsigned long long *__restrict__ mask, size_t mask_stride, int n_streams,
					   int n_blocks, long long sample_base, ChainState *__restrict__ states,
					   const ChainParams &p, int slot, tfrec_amd_event *__restrict__ events,
					   EventBuf *__restrict__ eb, uint32_t flags)
{
	const int s = blockIdx.x * 64 + threadIdx.x;
	if (s >= n_streams)
		return;
	ChainState &st = states[s];
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	const unsigned long long *mrow = mask + (size_t)s * mask_stride;
	const int m_total = n_blocks * kBlockDec;
	const int nwords = m_total >> 6;

	EmitCtx e{ events, eb, flags, (uint32_t)s, slot, p.sensor_type, sample_base };
	Dec d{ st.sr, st.sr_cnt, st.byte_cnt, st.invert, st.synced, st.w_last_bit, st.psk, st.last_psk, st.nrzs, st.lfsr, st.seq,
	       st.rdata };

	int lbi = st.last_bit_idx;
	int timeout_cnt = st.timeout_cnt;
	int mark_lvl = st.mark_lvl, rssi_i = st.rssi_i;
	int bitcnt = st.bitcnt, dmin = st.dmin, dmax = st.dmax, offset = st.offset, last_bit = st.last_bit;
	int last_dev = st.last_dev, avg_of = st.avg_of;
	unsigned long long step = st.step, last_peak = st.last_peak;
	double rssi_d = st.rssi_d;
	Biquad iir = st.iir, iir_avg = st.iir_avg;
	const double spb = p.spb;

	int g = 0;
	int cur_block = -1;  // demodulator::start has been applied for blocks <= cur_block
	int pg = -2;         // decimated index whose sample is in (pI, pQ)
	int pI = st.prev_i, pQ = st.prev_q;

	while (true) {
		if (timeout_cnt == 0) {
			// closed window: jump to the next sample with pwr > thresh (tfa1.cpp:147, tfa2.cpp:351, whb.cpp:636)
			int w = g >> 6;
			if (w >= nwords)
				break;
			unsigned long long m = mrow[w] & (~0ull << (g & 63));
			while (m == 0 && ++w < nwords)
				m = mrow[w];
			if (m == 0)
				break;
			g = (w << 6) + __builtin_ctzll(m);
		}
		if (g >= m_total)
			break;
		const int b = g >> 13;
		if (b != cur_block) {  // demodulator::start(len) once per block, decoder.cpp:118-122
			if (lbi)
				lbi -= kIndexSpan * (b - cur_block);
			cur_block = b;
		}
		const int index = 2 * (g & (kBlockDec - 1));
		if (pg != g - 1) {
			if (g > 0) {
				const uint32_t pw = drow[g - 1];
				pI = (int)(int16_t)(pw & 0xffff);
				pQ = (int)pw >> 16;
			} else {
				pI = st.prev_i;
				pQ = st.prev_q;
			}
		}
		const uint32_t cw = drow[g];
		const int I = (int)(int16_t)(cw & 0xffff), Q = (int)cw >> 16;
		const bool trig = (mrow[g >> 6] >> (g & 63)) & 1;

		if (KIND == 0) {
			// ---------------- tfa1_demod::demod, tfa1.cpp:143-190 (BITPERIOD 10)
			if (trig)
				timeout_cnt = p.window;
			// timeout_cnt != 0 here by construction
			const int dev = fm_dev_nrzs(I, Q, pI, pQ);
			if (dev > mark_lvl)
				mark_lvl = dev;
			else
				mark_lvl = d2i(mark_lvl * 0.95);
			if (mark_lvl > rssi_i)
				rssi_i = mark_lvl;
			timeout_cnt--;
			if (dev < mark_lvl / 2) {
				if (lbi) {
					const int gap = index - lbi;
					if (gap > 4) {
						for (int n = 22; n <= gap; n += 20)
							store_bit<0>(d, 1);
						store_bit<0>(d, 0);
					}
				}
				if (index - lbi > 2)
					lbi = index;
			}
			if (!timeout_cnt) {
				flush<0>(e, d, rssi_i, 0, g);
				mark_lvl = 0;
				rssi_i = 0;
				lbi = 0;
			}
		} else if (KIND == 1) {
			// ---------------- tfa2_demod::demod, tfa2.cpp:346-442
			if (trig) {
				if (!timeout_cnt) {  // tfa2_demod::reset, tfa2.cpp:325-334
					offset = 0;
					bitcnt = 0;
					dmin = 32767;
					dmax = -32767;
					last_bit = 0;
					rssi_i = 0;
				}
				timeout_cnt = p.window;
			}
			const int dev0 = fm_dev(I, Q, pI, pQ, eb, kAtanPolySerial);
			const int ld = d2i(iir_step(iir, p.iir, (double)dev0));
			if (bitcnt < 10) {
				if (ld > dmax)
					dmax = (7 * dmax + ld) / 8;
				if (ld < dmin)
					dmin = (7 * dmin + ld) / 8;
				offset = (dmax + dmin) / 2;
				if (bitcnt > 4) {  // wrapping int32 arithmetic as in the reference binary
					const uint32_t t = (uint32_t)rssi_i + (uint32_t)(I * I) + (uint32_t)(Q * Q);
					rssi_i = (int)((uint32_t)rssi_i + (uint32_t)((int)t / 100));
				}
			}
			timeout_cnt--;
			const int noffset = d2i(0.9 * offset);
			const int hi = noffset + dmax / 32, lo = noffset + dmin / 32;
			const int bit = ld > hi ? 1 : 0;
			if ((ld > hi || ld < lo) && bit != last_bit) {
				if (index > lbi + 8) {
					bitcnt++;
					const int tdiff = index - lbi;
					if (tdiff > spb / 4 && tdiff < 32 * spb) {
						const int bit_diff = tdiff / 2;
						const int numbits = d2i((bit_diff + (spb / 2)) / spb);
						if (numbits < 32)
							for (int n = 1; n < numbits; n++)
								store_bit<1>(d, last_bit);
						store_bit<1>(d, bit);
						last_bit = bit;
					}
				}
				if (index - lbi > 2)
					lbi = index;
			}
			if (!timeout_cnt) {
				for (int n = 0; n < 16; n++)
					store_bit<1>(d, last_bit);
				flush<1>(e, d, rssi_i, offset, g);
				offset = 0;
				bitcnt = 0;
				dmin = 32767;
				dmax = -32767;
				last_bit = 0;
				rssi_i = 0;
			}
		} else {
			// ---------------- whb_demod::demod, whb.cpp:632-707
			if (trig) {
				if (!timeout_cnt) {  // whb_demod::reset, whb.cpp:616-623
					rssi_d = 0;
					step = last_peak = 0;
				}
				timeout_cnt = p.window;
			}
			int dev = fm_dev_nrzs(I, Q, pI, pQ);
			dev = d2i(iir_step(iir, p.iir, (double)dev));
			if (!d.synced)
				avg_of = d2i(iir_step(iir_avg, p.iir_avg, 0.5 * dev));
			timeout_cnt--;
			const int tdiff = (int)(step - last_peak);
			if (dev < avg_of && dev > last_dev && (tdiff > 3 * spb / 4)) {
				store_bit<2>(d, 0);
				const int bit0 = d2i((tdiff + spb / 2) / spb);
				for (int n = 1; n < bit0; n++)
					store_bit<2>(d, 1);
				last_peak = step;
			}
			last_dev = dev;
			if (d.synced)
				rssi_d += (double)(I * I + Q * Q);
			if (!timeout_cnt) {
				if (d.synced) {
					for (int n = 0; n < 16; n++)
						store_bit<2>(d, 0);
					flush<2>(e, d, (long long)rssi_d, 0, g);
				}
				rssi_d = 0;
				step = last_peak = 0;
				// step++ below leaves 1, as the reference does; irrelevant: reset again at the next window
			}
			step++;
		}
		pI = I;
		pQ = Q;
		pg = g;
		g++;
	}
	// demodulator::start of the blocks this lane skipped at the end
	if (lbi)
		lbi -= kIndexSpan * (n_blocks - 1 - cur_block);
	// last decimated sample of the submit becomes last_i/last_q of the next one
	{
		const uint32_t lw = drow[m_total - 1];
		st.prev_i = (int)(int16_t)(lw & 0xffff);
		st.prev_q = (int)lw >> 16;
	}
	st.last_bit_idx = lbi;
	st.timeout_cnt = timeout_cnt;
	st.mark_lvl = mark_lvl;
	st.rssi_i = rssi_i;
	st.bitcnt = bitcnt;
	st.dmin = dmin;
	st.dmax = dmax;
	st.offset = offset;
	st.last_bit = last_bit;
	st.last_dev = last_dev;
	st.avg_of = avg_of;
	st.step = step;
	st.last_peak = last_peak;
	st.rssi_d = rssi_d;
	st.iir = iir;
	st.iir_avg = iir_avg;
	st.sr = d.sr;
	st.sr_cnt = d.sr_cnt;
	st.byte_cnt = d.byte_cnt;
	st.invert = d.invert;
	st.synced = d.synced;
	st.w_last_bit = d.w_last_bit;
	st.psk = d.psk;
	st.last_psk = d.last_psk;
	st.nrzs = d.nrzs;
	st.lfsr = d.lfsr;
	st.seq = d.seq;
}

// One launch covers every active slot: blockIdx.y selects the slot, so the five protocol chains of a
// stream run concurrently on different waves (each wave = 64 streams of one protocol).
__global__ __launch_bounds__(64) void chains_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						    const unsigned long long *__restrict__ mask, size_t mask_stride,
						    int n_streams, int n_blocks, long long sample_base, ChainLaunch L,
						    tfrec_amd_event *__restrict__ events, EventBuf *__restrict__ eb, uint32_t flags)
{
	const int a = blockIdx.y;
	const ChainParams &p = L.params[a];
	if (p.kind == 0)
		chain_body<0>(dec, dec_stride, mask, mask_stride, n_streams, n_blocks, sample_base, L.states[a], p, L.slot[a],
			      events, eb, flags);
	else if (p.kind == 1)
		chain_body<1>(dec, dec_stride, mask, mask_stride, n_streams, n_blocks, sample_base, L.states[a], p, L.slot[a],
			      events, eb, flags);
	else
		chain_body<2>(dec, dec_stride, mask, mask_stride, n_streams, n_blocks, sample_base, L.states[a], p, L.slot[a],
			      events, eb, flags);
}

hipError_t launch_chains(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			 size_t mask_stride, int n_streams, int n_blocks, long long sample_base, const ChainLaunch &L,
			 tfrec_amd_event *events, EventBuf *eb, uint32_t flags)
{
	if (L.n_active == 0)
		return hipSuccess;
	dim3 grid((n_streams + 63) / 64, L.n_active), block(64);
	hipLaunchKernelGGL(chains_kernel, grid, block, 0, st, dec, dec_stride, mask, mask_stride, n_streams, n_blocks,
			   sample_base, L, events, eb, flags);
	return hipGetLastError();
}

}  // namespace tfrec
