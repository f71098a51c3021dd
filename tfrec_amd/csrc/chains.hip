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
#include "tfrec_dev.h"

namespace tfrec {

// (int)double with x86 cvttsd2si semantics (NaN / out of range -> INT_MIN), SURVEY App. E.5
__device__ __forceinline__ int d2i(double v)
{
	if (!(v > -2147483649.0 && v < 2147483648.0))
		return (int)0x80000000;
	return (int)v;
}

__device__ __forceinline__ double iir_step(Biquad &f, const BiquadCoef &c, double dn)
{
	const double y1 = f.yn, y2 = f.yn1;
	const double y = ((c.b2 * f.dn2 + c.a1 * y1) + (c.b0 * dn + c.b1 * f.dn1)) + c.a2 * y2;
	f.yn1 = y1;
	f.yn = y;
	f.dn2 = f.dn1;
	f.dn1 = dn;
	return y;
}

// dsp_stuff.cpp:269-279
__device__ __forceinline__ int fm_dev_nrzs(int ar, int aj, int br, int bj)
{
	int cr = (int)((uint32_t)(ar * br) + (uint32_t)(aj * bj));
	cr = cr > 1000000000 ? 1000000000 : cr;
	cr = cr < -1000000000 ? -1000000000 : cr;
	return cr;
}

// dsp_stuff.cpp:284-292 in the arithmetic of the normative build: (int)(atan2(cj,cr) * (16384/pi)).
// Exactly representable directions (axes, diagonals, signed zeros) are resolved explicitly with the
// values glibc returns for them so they do not depend on the device atan2's last bit; everywhere else a
// 1-2 ulp difference can only matter when the product is within ~1e-11 of an integer, which is counted.
__device__ __forceinline__ int fm_dev(int ar, int aj, int br, int bj, unsigned long long *uncertain)
{
	const double cr = ((double)ar) * br + ((double)aj) * bj;
	const double cj = ((double)aj) * br - ((double)ar) * bj;
	const double kPi = 0x1.921fb54442d18p+1, kPi2 = 0x1.921fb54442d18p+0, kPi4 = 0x1.921fb54442d18p-1,
		     k3Pi4 = 0x1.2d97c7f3321d2p+1;
	const double kScale = 16384.0 * (1.0 / 0x1.921fb54442d18p+1);
	double ang;
	bool generic = false;
	if (cj == 0.0) {
		const bool pos = cr > 0.0 || (cr == 0.0 && !signbit(cr));
		ang = copysign(pos ? 0.0 : kPi, cj);
	} else if (cr == 0.0) {
		ang = copysign(kPi2, cj);
	} else if (fabs(cj) == fabs(cr)) {
		ang = copysign(cr > 0.0 ? kPi4 : k3Pi4, cj);
	} else {
		ang = atan2(cj, cr);
		generic = true;
	}
	const double v = ang * kScale;
	if (generic) {
		const double fr = fabs(v - rint(v));
		if (fr < 1e-9)
			atomicAdd(uncertain, 1ull);
	}
	return d2i(v);
}

// CRC-8 poly 0x31 init 0 MSB-first (crc8.cpp:4-28), bitwise
__device__ inline uint8_t crc8_31(const uint8_t *d, int n)
{
	uint32_t c = 0;
	for (int i = 0; i < n; i++) {
		c ^= d[i];
		for (int b = 0; b < 8; b++)
			c = (c & 0x80) ? ((c << 1) ^ 0x31) & 0xff : (c << 1) & 0xff;
	}
	return (uint8_t)c;
}

// CRC-32 poly 0x04c11db7 MSB-first, caller init, no reflection / xorout (crc32.cpp:4-30), bitwise
__device__ inline uint32_t crc32_04c11db7(const uint8_t *d, int n, uint32_t c)
{
	for (int i = 0; i < n; i++) {
		c ^= (uint32_t)d[i] << 24;
		for (int b = 0; b < 8; b++)
			c = (c & 0x80000000u) ? ((c << 1) ^ 0x04c11db7u) : (c << 1);
	}
	return c;
}

// crc_initvals (whb.cpp:50-62)
__device__ inline bool whb_crc_init(uint32_t stype, uint32_t *init)
{
	switch (stype) {
	case 0x02: *init = 0x97d97a26u; return true;
	case 0x03: *init = 0xf59c5a1eu; return true;
	case 0x04: *init = 0x98e1d11fu; return true;
	case 0x06: *init = 0xa7a41254u; return true;
	case 0x07: *init = 0x3303fb1du; return true;
	case 0x08: *init = 0x29f0f49bu; return true;
	case 0x09: *init = 0xa7a41254u; return true;
	case 0x0b: *init = 0xe7720ae4u; return true;
	case 0x10: *init = 0x62d0afc1u; return true;
	case 0x11: *init = 0x8cba0708u; return true;
	case 0x12: *init = 0x5a9e30aeu; return true;
	default: return false;
	}
}

// Would the reference decoder's flush() accept what is in rdata?  1 = telegram, 2 = rejected, 0 = too short.
template <int KIND>
__device__ inline int flush_verdict(const uint8_t *r, int byte_cnt, int sensor_type)
{
	if (KIND == 0) {  // tfa1.cpp:49-73
		if (byte_cnt < 10)
			return 0;
		const int hum = r[6];
		const bool ok = r[10] == crc8_31(r + 2, 8) && ((r[4] & 0xf0) == 0x80 || hum == 0x7f || hum == 0x6a) &&
				hum <= 0x7f && (r[7] & 0x60) == 0x60 && (r[8] & 0xf) == 0 && r[9] == 0x56;
		return ok ? 1 : 2;
	} else if (KIND == 1) {
		if (sensor_type == 3) {  // TX22, tfa2.cpp:76-93
			if (byte_cnt < 7 || byte_cnt >= 64)
				return 0;
			if ((r[2] >> 4) != 0xa)
				return 2;
			const int num = r[3] & 7;
			return r[2 * num + 4] == crc8_31(r + 2, 2 + 2 * num) ? 1 : 2;
		}
		if (byte_cnt < 7)  // tfa2.cpp:222-237
			return 0;
		return r[6] == crc8_31(r + 2, 4) ? 1 : 2;
	} else {  // whb.cpp:484-510
		if (byte_cnt < 11 || byte_cnt > 60)
			return 0;
		const int plen = r[4];
		uint32_t init;
		if (plen > 60 || !whb_crc_init(r[5], &init))
			return 2;
		const uint32_t calc = crc32_04c11db7(r + 4, plen - 4, init);
		const uint32_t val = ((uint32_t)r[plen] << 24) | ((uint32_t)r[plen + 1] << 16) | ((uint32_t)r[plen + 2] << 8) |
				     r[plen + 3];
		return calc == val ? 1 : 2;
	}
}

struct EmitCtx {
	tfrec_amd_event *events;
	EventBuf *eb;
	uint32_t flags;
	uint32_t stream;
	int slot;
	int sensor_type;
	long long sample_base;
};

// one lane's working copy of the decoder (registers) + its rdata in global memory
struct Dec {
	uint32_t sr;
	int sr_cnt, byte_cnt, invert, synced;
	int w_last_bit, psk, last_psk, nrzs;
	uint32_t lfsr;
	uint32_t seq;
	uint8_t *rdata;
};

template <int KIND>
__device__ __forceinline__ void store_bit(Dec &d, int bit)
{
	if (KIND == 0) {  // tfa1.cpp:120-134, LSB first, sync 0xd42d in the oldest 16 bits
		d.sr = (d.sr >> 1) | ((uint32_t)bit << 31);
		if ((d.sr & 0xffff) == 0xd42d) {
			d.sr_cnt = 0;
			d.byte_cnt = 0;
		}
		if (d.sr_cnt == 0) {
			if (d.byte_cnt < 256)
				d.rdata[d.byte_cnt] = d.sr & 0xff;
			d.byte_cnt++;
		}
	} else if (KIND == 1) {  // tfa2.cpp:281-314, MSB first, sync 0x2dd4 or its complement
		d.sr = (d.sr << 1) | (uint32_t)bit;
		if ((d.sr & 0xffff) == 0x2dd4) {
			d.sr_cnt = 0;
			d.rdata[0] = (d.sr >> 8) & 0xff;
			d.byte_cnt = 1;
			d.invert = 0;
		}
		if (((~d.sr) & 0xffff) == 0x2dd4) {
			d.sr_cnt = 0;
			d.rdata[0] = (uint8_t) ~((d.sr >> 8) & 0xff);
			d.byte_cnt = 1;
			d.invert = 1;
		}
		if (d.sr_cnt == 0) {
			if (d.byte_cnt < 256)
				d.rdata[d.byte_cnt] = d.invert ? (uint8_t) ~(d.sr & 0xff) : (uint8_t)(d.sr & 0xff);
			d.byte_cnt++;
		}
	} else {  // whb.cpp:566-603: de-PSK, de-NRZS, G3RUH descrambler, LSB first, 32-bit sync
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
		}
		if (d.sr_cnt == 0) {
			if (d.byte_cnt < 256)
				d.rdata[d.byte_cnt] = (d.sr >> 24) & 0xff;
			d.byte_cnt++;
		}
	}
	if (d.sr_cnt >= 0)
		d.sr_cnt = (d.sr_cnt + 1) & 7;
}

// decoder::flush as seen from the demodulator: report, then the decoder's own resets
// (tfa1.cpp:115-117, tfa2.cpp:213-216/276-278, whb.cpp:559-563).
template <int KIND>
__device__ inline void flush(const EmitCtx &e, Dec &d, long long rssi_raw, int offset, int g)
{
	const int verdict = flush_verdict<KIND>(d.rdata, d.byte_cnt, e.sensor_type);
	if ((e.flags & TFREC_AMD_F_ALL_FLUSHES) || verdict != 0) {
		const uint32_t idx = atomicAdd(&e.eb->count, 1u);
		if (idx < e.eb->capacity) {
			tfrec_amd_event *ev = e.events + idx;
			ev->stream = e.stream;
			ev->slot = (uint8_t)e.slot;
			ev->status = (uint8_t)verdict;
			ev->byte_cnt = (uint16_t)(d.byte_cnt > 65535 ? 65535 : d.byte_cnt);
			ev->offset = offset;
			ev->seq = d.seq;
			ev->end_sample = e.sample_base + g;
			ev->rssi_raw = rssi_raw;
			const uint4 *src = reinterpret_cast<const uint4 *>(d.rdata);
			uint4 *dst = reinterpret_cast<uint4 *>(ev->rdata);
			dst[0] = src[0];
			dst[1] = src[1];
			dst[2] = src[2];
			dst[3] = src[3];
		}
	}
	d.seq++;
	d.sr_cnt = -1;
	d.byte_cnt = 0;
	if (KIND == 0) {
		d.rdata[10] = 0;
	} else {
		d.sr = 0;
		if (KIND == 2)
			d.synced = 0;
	}
}

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
			const int dev0 = fm_dev(I, Q, pI, pQ, &eb->uncertain);
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
