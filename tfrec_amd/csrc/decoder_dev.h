// tfrec_amd/csrc/decoder_dev.h -- device-side decoders: store_bit (sync search + byte packing), the
// acceptance tests of flush() and event emission.  Shared by the serial reference chains (chains.hip)
// and the window-parallel pipeline (chains2.hip).
//   tfa1_decoder::store_bit tfa1.cpp:120-134   flush tfa1.cpp:47-118
//   tfa2_decoder::store_bit tfa2.cpp:281-314   flush_tfa tfa2.cpp:219-279, flush_tx22 tfa2.cpp:72-217
//   whb_decoder::store_bit  whb.cpp:566-603    flush whb.cpp:477-564, crc_initvals whb.cpp:50-62
//   crc8::calc crc8.cpp:19-28, crc32::calc crc32.cpp:19-30
#pragma once

#include "dsp_dev.h"

namespace tfrec {

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
	bool quiet = false;  // compute everything, report nothing (the non-leading lanes of a wave-per-chain kernel)
};

// TFREC_AMD_F_BITS: `nbits` bits (LSB first in words[]) that precede flush number `seq` of this (stream, slot), as chunks
// of up to 512; `g_open` (the window's first sample) places them among the other bits of the same flush, the chunk index
// (from chunk0 on) goes to `offset`
__device__ inline void emit_bits(const EmitCtx &e, uint32_t seq, int g_open, int chunk0, const uint32_t *words, int nbits)
{
	if (e.quiet)
		return;
	for (int b0 = 0, chunk = chunk0; b0 < nbits; b0 += 512, chunk++) {
		const int n = nbits - b0 < 512 ? nbits - b0 : 512;
		const uint32_t idx = atomicAdd(&e.eb->count, 1u);
		if (idx >= e.eb->capacity)
			continue;
		tfrec_amd_event *ev = e.events + idx;
		ev->stream = e.stream;
		ev->slot = (uint8_t)e.slot;
		ev->status = (uint8_t)TFREC_AMD_STATUS_BITS;
		ev->byte_cnt = (uint16_t)n;
		ev->offset = chunk;
		ev->seq = seq;
		ev->end_sample = e.sample_base + g_open;
		ev->rssi_raw = 0;
		uint32_t *dst = reinterpret_cast<uint32_t *>(ev->rdata);
		for (int w = 0; w < 16; w++) {
			const int bw = b0 + 32 * w;
			uint32_t v = bw < nbits ? words[bw >> 5] : 0u;
			if (bw < nbits && nbits - bw < 32)
				v &= (1u << (nbits - bw)) - 1u;
			dst[w] = v;
		}
	}
}

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
// Returns the index of the event it appended (-1: none).
template <int KIND>
__device__ inline int flush(const EmitCtx &e, Dec &d, long long rssi_raw, int offset, int g)
{
	const int verdict = flush_verdict<KIND>(d.rdata, d.byte_cnt, e.sensor_type);
	int ev_idx = -1;
	if (!e.quiet && ((e.flags & TFREC_AMD_F_ALL_FLUSHES) || verdict != 0)) {
		const uint32_t idx = atomicAdd(&e.eb->count, 1u);
		if (idx < e.eb->capacity) {
			ev_idx = (int)idx;
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
	return ev_idx;
}

}  // namespace tfrec
