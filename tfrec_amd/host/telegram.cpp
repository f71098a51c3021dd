// tfrec_amd/host/telegram.cpp -- host side of the hot path's tail: result store + telegram parsers.
//
// Mirrors the behaviour of the reference's decoder.cpp:46-109 (store_data / execute_handler / flush_storage),
// tfa1.cpp:47-134, tfa2.cpp:64-314, whb.cpp:109-603 and crc8.cpp / crc32.cpp, written from the on-air formats
// and pinned against the real reference by tests/golden/kat_bytes.json (tests/test_host_cpp.py).
//
// Numeric note: the reference's normative build (-O3 -ffast-math) turns every "/10" of the field maths into
// "*0.1"; the doubles stored in sensordata_t follow that arithmetic (see oracle/tfrec_oracle.c header).
#include "plugin.h"

#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------- decoder base (decoder.cpp:9-109)
decoder::decoder(sensor_e _type)
	: dbg(0), bad(0), synced(0), type(_type), byte_cnt(0), handler(NULL), mode(0)
{
	memset(rdata, 0, sizeof(rdata));
}

void decoder::set_params(char *_handler, int _mode, int _dbg)
{
	handler = _handler;
	mode = _mode;
	dbg = _dbg;
}

void decoder::store_bit(int) {}
void decoder::flush(int, int) {}

// "Shortcut for testing" in the reference (decoder.cpp:35-40): the entry point the GPU adapter uses
void decoder::store_bytes(uint8_t *d, int len)
{
	if (len > (int)sizeof(rdata))
		len = sizeof(rdata);
	memcpy(rdata, d, len);
	byte_cnt = len;
	synced = 1;
}

// first appearance of an id is kept; WHB repeats with an unchanged sequence are not executed twice
void decoder::store_data(sensordata_t &d)
{
	// test hook (not in the reference): TFREC_HOST_RECORDS=1 prints every record exactly
	static const bool dump = getenv("TFREC_HOST_RECORDS") != NULL;
	if (dump)
		printf("D %i %" PRIx64 " %a %a %i %i %i %i\n", (int)d.type, d.id, d.temp, d.humidity, d.sequence, d.alarm, d.rssi,
		       d.flags);
	bool repeat = false;
	std::map<uint64_t, sensordata_t>::iterator it = data.find(d.id);
	if (it == data.end())
		data.insert(std::make_pair(d.id, d));
	else if (it->second.type == TFA_WHB) {
		if (it->second.sequence == d.sequence)
			repeat = true;
		else
			it->second.sequence = d.sequence;
	}
	if (!mode && !repeat)
		execute_handler(d);
}

// handler command line: cmd id temp hum seq alarm rssi flags ts (decoder.cpp:67-96)
void decoder::execute_handler(sensordata_t &d)
{
	if (!handler || !*handler)
		return;
	char args[384];
	if (type != TFA_WHB)
		snprintf(args, sizeof(args), "%04" PRIx64 " %+.1f %g %i %i %i %i %li", d.id | ((uint64_t)d.type << 24), d.temp,
			 d.humidity, d.sequence, d.alarm, d.rssi, d.flags, (long)d.ts);
	else
		snprintf(args, sizeof(args), "%013" PRIx64 " %+.1f %g %i %i %i %i %li", d.id, d.temp, d.humidity, d.sequence, d.alarm,
			 d.rssi, d.flags, (long)d.ts);
	char cmd[512];
	snprintf(cmd, sizeof(cmd), "%s %s", handler, args);
	if (dbg >= 1)
		printf("EXEC %s\n", cmd);
	if (system(cmd)) {
	}
}

void decoder::flush_storage(void)
{
	if (!mode)
		return;
	for (std::map<uint64_t, sensordata_t>::iterator it = data.begin(); it != data.end(); ++it)
		execute_handler(it->second);
	data.clear();
}

demodulator::demodulator(decoder *_dec) : dec(_dec), last_bit_idx(0) {}
void demodulator::start(int len)
{
	if (last_bit_idx)
		last_bit_idx -= len;
}
int demodulator::demod(int, int, int, int16_t *) { return 0; }

// ---------------------------------------------------------------- CRCs (MSB first, table driven)
crc8::crc8(int poly)
{
	for (int n = 0; n < 256; n++) {
		uint8_t t = (uint8_t)n;
		for (int m = 0; m < 8; m++)
			t = (t & 0x80) ? (uint8_t)((t << 1) ^ poly) : (uint8_t)(t << 1);
		lookup[n] = t;
	}
}
uint8_t crc8::calc(uint8_t *p, int len)
{
	uint8_t c = 0;
	while (len-- > 0)
		c = lookup[c ^ *p++];
	return c;
}
crc32::crc32(uint32_t poly)
{
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t t = n << 24;
		for (int m = 0; m < 8; m++)
			t = (t & 0x80000000u) ? ((t << 1) ^ poly) : (t << 1);
		lookup[n] = t;
	}
}
uint32_t crc32::calc(uint8_t *p, int len, uint32_t init)
{
	uint32_t c = init;
	while (len-- > 0)
		c = (c << 8) ^ lookup[(c >> 24) ^ *p++];
	return c;
}

static sensordata_t make_sd(sensor_e type, uint64_t id, double temp, double hum, int seq, int alarm, int rssi)
{
	sensordata_t sd;
	sd.type = type;
	sd.id = id;
	sd.temp = temp;
	sd.humidity = hum;
	sd.sequence = seq;
	sd.alarm = alarm;
	sd.rssi = rssi;
	sd.flags = 0;
	sd.ts = time(0);
	return sd;
}

static inline int bcd3(int hundreds, uint8_t lo) { return hundreds * 100 + (lo >> 4) * 10 + (lo & 0xf); }

// ---------------------------------------------------------------- TFA_1 (KlimaLogg Pro): 2d d4 ID ID sT TT HH BB SS 56 CC
tfa1_decoder::tfa1_decoder(sensor_e _type) : decoder(_type), sr(0), sr_cnt(-1), snum(0), crc(0x31) { byte_cnt = 0; }

// debug mode (-D): every candidate telegram's running number, wall-clock second and first bytes ahead of the verdict
// (tfa1.cpp:50-55)
void tfa1_decoder::debug_header(int nbytes, const char *tail)
{
	printf("#%03i %u  ", snum++, (uint32_t)time(0));
	for (int n = 0; n < nbytes; n++)
		printf("%02x ", rdata[n]);
	fputs(tail, stdout);
}

void tfa1_decoder::store_bit(int bit)  // LSB-first shift register, sync d4 2d in the oldest 16 bits
{
	sr = (sr >> 1) | ((uint32_t)bit << 31);
	if ((sr & 0xffff) == 0xd42d)
		sr_cnt = byte_cnt = 0;
	if (sr_cnt == 0) {
		if (byte_cnt < (int)sizeof(rdata))
			rdata[byte_cnt] = sr & 0xff;
		byte_cnt++;
	}
	if (sr_cnt >= 0)
		sr_cnt = (sr_cnt + 1) & 7;
}

void tfa1_decoder::flush(int rssi, int)
{
	const uint8_t *r = rdata;
	if (byte_cnt >= 10) {
		const int id = ((r[2] << 8) | r[3]) & 0x7fff;
		int lowbat = r[7] >> 7;
		double temp = bcd3(r[4] & 0xf, r[5]) * 0.1 - 40;
		int hum = r[6];
		const uint8_t want = crc.calc(&rdata[2], 8);
		const bool learning_ok = (r[4] & 0xf0) == 0x80 || hum == 0x7f || hum == 0x6a;
		if (dbg)
			debug_header(11, "          ");
		if (r[10] == want && learning_ok && hum <= 0x7f && (r[7] & 0x60) == 0x60 && (r[8] & 0xf) == 0 && r[9] == 0x56) {
			if (hum == 0x6a)  // temperature-only sensors
				hum = 0;
			if (r[5] == 0xff || r[5] == 0xaa || hum == 0x7f) {  // sensor values invalid (low battery)
				lowbat = 2;
				hum = 0;
				temp = 0;
			}
			if (dbg >= 0) {
				printf("TFA1 ID %04x %+.1f %i%% seq %x lowbat %i RSSI %i\n", id, temp, hum, r[8] >> 4, lowbat, rssi);
				fflush(stdout);
			}
			sensordata_t sd = make_sd(TFA_1, id, temp, hum, r[8] >> 4, lowbat, rssi);
			store_data(sd);
		} else {
			bad++;
			if (dbg) {
				if (r[10] != want)
					printf("TFA1 BAD %i RSSI %i (CRC %02x %02x)\n", bad, rssi, r[10], want);
				else
					printf("TFA1 BAD %i RSSI %i (SANITY)\n", bad, rssi);
			}
		}
	}
	sr_cnt = -1;
	byte_cnt = 0;
	rdata[10] = 0;
}

// ---------------------------------------------------------------- TFA_2 / TFA_3 / TX22
tfa2_decoder::tfa2_decoder(sensor_e _type) : decoder(_type), invert(0), sr(0), sr_cnt(-1), snum(0), crc(0x31) { byte_cnt = 0; }

// (tfa2.cpp:77-82 with the telegram's byte count, :152-157 with seven bytes)
void tfa2_decoder::debug_header(int nbytes, const char *tail)
{
	printf("#%03i %u  ", snum++, (uint32_t)time(0));
	for (int n = 0; n < nbytes; n++)
		printf("%02x ", rdata[n]);
	fputs(tail, stdout);
}

void tfa2_decoder::store_bit(int bit)  // MSB-first shift register, sync 2d d4 (or its complement)
{
	sr = (sr << 1) | (uint32_t)bit;
	const uint32_t lo = sr & 0xffff;
	if (lo == 0x2dd4 || lo == (uint16_t)~0x2dd4) {
		invert = lo != 0x2dd4;
		if (invert)
			printf("Inverted SYNC\n");
		sr_cnt = 0;
		rdata[0] = invert ? (uint8_t) ~(sr >> 8) : (uint8_t)(sr >> 8);
		byte_cnt = 1;
	}
	if (sr_cnt == 0) {
		if (byte_cnt < (int)sizeof(rdata))
			rdata[byte_cnt] = invert ? (uint8_t)~sr : (uint8_t)sr;
		byte_cnt++;
	}
	if (sr_cnt >= 0)
		sr_cnt = (sr_cnt + 1) & 7;
}

void tfa2_decoder::rearm()
{
	sr_cnt = -1;
	sr = 0;
	byte_cnt = 0;
}

void tfa2_decoder::flush(int rssi, int offset)
{
	if (type == TX22)
		flush_tx22(rssi, offset);
	else
		flush_tfa(rssi, offset);
}

// 2d d4 II IT TT HH CC
void tfa2_decoder::flush_tfa(int rssi, int offset)
{
	const uint8_t *r = rdata;
	if (byte_cnt >= 7) {
		if (dbg)
			debug_header(7, "                      ");
		int id = (type << 28) | (r[2] << 8) | (r[3] & 0xc0);
		const double temp = bcd3(r[3] & 0xf, r[4]) * 0.1 - 40;
		int hum = r[5];
		if (hum == 0x7d)  // external temperature probe -> sub-id 1
			id |= 1;
		const uint8_t want = crc.calc(&rdata[2], 4);
		if (r[6] == want) {
			if (hum > 100)
				hum = 0;
			if (dbg >= 0) {
				printf("TFA%i ID %06x %+.1lf %i%% RSSI %i Offset %.0lfkHz\n", type + 1, id, temp, hum, rssi,
				       -1536.0 * offset / 131072);
				fflush(stdout);
			}
			sensordata_t sd = make_sd(type, (uint64_t)(int64_t)id, temp, hum, 0, 0, rssi);
			store_data(sd);
		} else {
			bad++;
			if (dbg)
				printf("TFA%i BAD %i RSSI %i  Offset %.0lfkHz (CRC %02x %02x)\n", type + 1, bad, rssi,
				       -1536.0 * offset / 131072, r[6], want);
		}
	}
	rearm();
}

// 2d d4 SI IQ (TV VV)*n CC : typed 12-bit words
void tfa2_decoder::flush_tx22(int rssi, int offset)
{
	const uint8_t *r = rdata;
	const bool in_range = byte_cnt >= 7 && byte_cnt < 64;
	uint8_t got = 0, want = 0;  // (both zero where the telegram fails before its CRC is looked at: "SANITY")
	bool ok = false;
	if (in_range && dbg)
		debug_header(byte_cnt, "      ");
	if (in_range && (r[2] >> 4) == 0xa) {
		const int num = r[3] & 7;
		got = r[2 * num + 4];
		want = crc.calc(&rdata[2], 2 + 2 * num);
		if (got == want) {
			ok = true;
			const int sid = ((r[2] & 0xf) << 2) | (r[3] >> 6);
			const int alarm = (!((r[3] >> 4) & 1)) | ((r[3] >> 3) & 1);  // error | low battery
			bool have[5] = { false, false, false, false, false };
			double temp = 0, hum = 0, rain = 0, wdir = 0, wspeed = 0, gust = 0;
			for (int n = 0; n < num; n++) {
				const uint8_t *w = &r[4 + 2 * n];
				const int kind = w[0] >> 4, v12 = ((w[0] & 0xf) << 8) | w[1];
				if (kind > 4)
					continue;
				have[kind] = true;
				switch (kind) {
				case 0: temp = bcd3(w[0] & 0xf, w[1]) * 0.1 - 40; break;
				case 1: hum = bcd3(w[0] & 0xf, w[1]); break;
				case 2: rain = v12; break;
				case 3: wdir = (w[0] & 0xf) * 22.5; wspeed = w[1] * 0.1; break;
				case 4: gust = v12 * 0.1; break;
				}
			}
			const int base = (type << 28) | (sid << 4);
			if (dbg >= 0) {
				printf("TX22 ID %x, ", base);
				if (have[0]) printf("temp %g, ", temp);
				if (have[1]) printf("hum %g, ", hum);
				if (have[2]) printf("rain %g, ", rain);
				if (have[3]) printf("speed %g, dir %g, ", wspeed, wdir);
				if (have[4]) printf("gust %g, ", gust);
				printf("RSSI %i, offset %.0lfkHz\n", rssi, -1536.0 * offset / 131072);
				fflush(stdout);
			}
			const struct { bool on; int sub; double t, h; } out[4] = {
				{ have[0], 0, temp, hum }, { have[2], 2, rain, 0 }, { have[3], 3, wspeed, wdir }, { have[4], 4, gust, 0 } };
			for (int k = 0; k < 4; k++)
				if (out[k].on) {
					sensordata_t sd = make_sd(type, (uint64_t)(int64_t)(base | out[k].sub), out[k].t, out[k].h, 0, alarm, rssi);
					store_data(sd);
				}
		}
	}
	if (!ok && dbg && in_range) {  // (counted in debug mode only: tfa2.cpp:133-134)
		bad++;
		if (got != want)
			printf("TX22(%02x) BAD %i RSSI %i  Offset %.0lfkHz (CRC %02x %02x) len %i\n", 1 << type, bad, rssi,
			       -1536.0 * offset / 131072, got, want, byte_cnt);
		else
			printf("TX22(%02x) BAD %i RSSI %i  Offset %.0lfkHz len %i (SANITY)\n", 1 << type, bad, rssi,
			       -1536.0 * offset / 131072, byte_cnt);
		fflush(stdout);
	}
	rearm();
}

// ---------------------------------------------------------------- WeatherHub: 4b 2d d4 2b LL ID*6 payload CRC32
whb_decoder::whb_decoder(sensor_e _type) : decoder(_type), sr(0), sr_cnt(-1), snum(0), crc(0x04c11db7), raw_hist(0) { byte_cnt = 0; }

void whb_decoder::store_bit(int bit)
{
	// de-PSK + de-NRZS + G3RUH descrambling of the reference collapse to a 3-tap XOR over the raw bits
	raw_hist = (raw_hist << 1) | (uint32_t)bit;
	const uint32_t out = (raw_hist ^ (raw_hist >> 12) ^ (raw_hist >> 17)) & 1;
	sr = (sr >> 1) | (out << 31);
	if (sr == 0x2bd42d4bu) {
		synced = 1;
		sr_cnt = 0;
		rdata[0] = 0x4b;
		rdata[1] = 0x2d;
		rdata[2] = 0xd4;
		byte_cnt = 3;
	}
	if (sr_cnt == 0) {
		if (byte_cnt < (int)sizeof(rdata))
			rdata[byte_cnt] = sr >> 24;
		byte_cnt++;
	}
	if (sr_cnt >= 0)
		sr_cnt = (sr_cnt + 1) & 7;
}

static inline unsigned be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// 11-bit (or 12-bit "extended") two's complement tenths of a degree
static double whb_temp(unsigned raw, bool ext = false)
{
	const unsigned sign = ext ? 0x800 : 0x400, full = ext ? 0xfff : 0x7ff;
	return (raw & sign) ? -(int)((raw ^ full) + 1) * 0.1 : raw * 0.1;
}

static const uint32_t k_unit_seconds[4] = { 86400, 3600, 60, 1 };

static bool whb_crc_init(uint32_t stype, uint32_t *init)
{
	static const struct { uint8_t t; uint32_t v; } tab[] = {
		{ 0x02, 0x97d97a26 }, { 0x03, 0xf59c5a1e }, { 0x04, 0x98e1d11f }, { 0x06, 0xa7a41254 }, { 0x07, 0x3303fb1d },
		{ 0x08, 0x29f0f49b }, { 0x09, 0xa7a41254 }, { 0x0b, 0xe7720ae4 }, { 0x10, 0x62d0afc1 }, { 0x11, 0x8cba0708 },
		{ 0x12, 0x5a9e30ae } };
	for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); k++)
		if (tab[k].t == stype) {
			*init = tab[k].v;
			return true;
		}
	return false;
}

void whb_decoder::emit(uint64_t id, int sub, double temp, double hum, int seq, int rssi)
{
	sensordata_t sd = make_sd(type, (id << 4) + sub, temp, hum, seq, 0, rssi);
	store_data(sd);
}

void whb_decoder::payload(uint32_t stype, const uint8_t *m, uint64_t id, int rssi)
{
	const int seq = be16(m) & 0x3fff;
	const bool show = dbg >= 0;
#define T11(off) whb_temp(be16(m + (off)) & 0x7ff)
#define H8(off) (int)(be16(m + (off)) & 0xff)
	switch (stype) {
	case 0x02:
		if (show) printf("WHB02 ID %" PRIx64 " TEMP %g, PTEMP %g\n", id, T11(2), T11(4));
		emit(id, 0, T11(2), 0, seq, rssi);
		break;
	case 0x03:
		if (show) printf("WHB03 ID %" PRIx64 " TEMP %g HUM %i, PTEMP %g PHUM %i\n", id, T11(2), H8(4), T11(6), H8(8));
		emit(id, 0, T11(2), H8(4), seq, rssi);
		break;
	case 0x04: {
		const int dry = (m[6] & 1) ^ 1, pdry = (m[11] & 1) ^ 1;
		if (show)
			printf("WHB04 ID %" PRIx64 " TEMP %g HUM %i WET %i, PTEMP %g PHUM %i PWET %i\n", id, T11(2), H8(4), dry, T11(7),
			       H8(9), pdry);
		emit(id, 0, T11(2), H8(4), seq, rssi);
		emit(id, 5, dry, 0, seq, rssi);
		break;
	}
	case 0x06:
	case 0x09: {
		const bool ext = stype == 0x09;
		const unsigned m2 = ext ? 0xfff : 0x7ff;
		const double t2 = whb_temp(be16(m + 4) & m2, ext), pt2 = whb_temp(be16(m + 10) & m2, ext);
		if (show)
			printf("WHB0%i ID %" PRIx64 "TEMP %g HUM %i TEMP2 %g, PTEMP %g PHUM %i PTEMP2 %g\n", ext ? 9 : 6, id, T11(2), H8(6),
			       t2, T11(8), H8(12), pt2);
		emit(id, 0, T11(2), H8(6), seq, rssi);
		emit(id, 1, t2, 0, seq, rssi);
		break;
	}
	case 0x07:
		if (show) {
			printf("WHB07 ID %" PRIx64 " TEMP_IN %g HUM_IN %i TEMP_OUT %g HUM_OUT %i", id, T11(2), H8(4), T11(6), H8(8));
			if (dbg > 1)
				printf(" PTEMP_IN %g PHUM_IN %i PTEMP_OUT %g PHUM_OUT %i", T11(10), H8(12), T11(14), H8(16));
			puts("");
		}
		emit(id, 0, T11(2), H8(4), seq, rssi);
		emit(id, 0xc, T11(6), H8(8), seq, rssi);
		break;
	case 0x08: {
		const unsigned cnt = be16(m + 4), x1 = be16(m + 8);
		if (show) printf("WHB08 ID %" PRIx64 " cnt %i\n", id, cnt);
		for (int i = 0; i < 10 && dbg > 1; i++) {  // the ten event times (-D -D)
			const unsigned x = be16(m + 6 + 2 * i);
			printf("WHB08 ID %" PRIx64 " #%i time %i\n", id, i, k_unit_seconds[(x >> 14) & 3] * (x & 0x3fff));
		}
		emit(id, 2, cnt, k_unit_seconds[(x1 >> 14) & 3] * (x1 & 0x3fff), seq, rssi);
		emit(id, 0, T11(2), 0, seq, rssi);
		break;
	}
	case 0x0b: {  // wind: 24-bit sequence, values kept in single precision like the reference
		const int seq24 = (m[0] << 16) | (m[1] << 8) | m[2];
		// the newest of six history entries (entry 0) is the reading; -D prints them all
		const uint32_t v0 = ((uint32_t)m[3] << 24) | (m[4] << 16) | (m[5] << 8) | m[6];
		const float dir = 22.5 * (v0 >> 28), speed = (((v0 >> 16) & 0xff) + 256 * ((v0 >> 25) & 1)) / 10.0,
			    gust = (((v0 >> 8) & 0xff) + 256 * ((v0 >> 24) & 1)) / 10.0;
		for (int i = 0; i < 6 && show && (i == 0 || dbg > 0); i++) {
			const uint8_t *e = m + 3 + 4 * i;
			const uint32_t v = ((uint32_t)e[0] << 24) | (e[1] << 16) | (e[2] << 8) | e[3];
			const float d = 22.5 * (v >> 28), sp = (((v >> 16) & 0xff) + 256 * ((v >> 25) & 1)) / 10.0,
				    gu = (((v >> 8) & 0xff) + 256 * ((v >> 24) & 1)) / 10.0;
			printf("WHB0b ID %" PRIx64 " #%i DIR %f SPEED %f GUST %f time %i\n", id, i, d, sp, gu, (v & 0xff) * 2);
		}
		emit(id, 3, speed, dir, seq24, rssi);
		emit(id, 4, gust, 0, seq24, rssi);
		break;
	}
	case 0x10: {
		const unsigned x0 = be16(m + 2), x1 = be16(m + 4);
		for (int i = 0; i < 4 && show && (i == 0 || dbg > 0); i++) {
			const unsigned x = be16(m + 2 + 2 * i);
			printf("WHB10 ID %" PRIx64 " #%i %i %i\n", id, i, x >> 15, k_unit_seconds[(x >> 13) & 3] * (x & 0x1fff));
		}
		emit(id, 5, x0 >> 15, k_unit_seconds[(x1 >> 13) & 3] * (x1 & 0x1fff), seq, rssi);
		break;
	}
	case 0x11:
		if (show) {
			printf("WHB11 %" PRIx64 " TEMP1 %g HUM1 %i TEMP2 %g HUM2 %i TEMP3 %g HUM3 %i TEMP_IN %g HUM_IN %i", id, T11(2), H8(4),
			       T11(6), H8(8), T11(10), H8(12), T11(14), H8(16));
			if (dbg > 1)
				printf(" PTEMP1 %g PHUM1 %i PTEMP2 %g PHUM2 %i PTEMP3 %g PHUM3 %i PTEMP_IN %g PHUM_IN %i", T11(18), H8(20), T11(22),
				       H8(24), T11(26), H8(28), T11(30), H8(32));
			puts("");
		}
		emit(id, 0, T11(14), H8(16), seq, rssi);
		for (int n = 0; n < 3; n++)
			emit(id, 0xc + n, T11(2 + 4 * n), H8(4 + 4 * n), seq, rssi);
		break;
	case 0x12: {
		const int h[5] = { m[8] & 0x7f, m[2] & 0x7f, m[3] & 0x7f, m[4] & 0x7f, m[5] & 0x7f };
		if (show)
			printf("WHB12 %" PRIx64 " TEMP %g HUM %i HUM3h %i HUM24h %i HUM7d %i HUM30d %i\n", id, T11(6), h[0], h[1], h[2], h[3],
			       h[4]);
		emit(id, 0, T11(6), h[0], seq, rssi);
		emit(id, 1, 0, h[1], seq, rssi);
		for (int n = 0; n < 3; n++)
			emit(id, 0xc + n, 0, h[2 + n], seq, rssi);
		break;
	}
	}
#undef T11
#undef H8
	if (show)
		fflush(stdout);
}

void whb_decoder::flush(int rssi, int)
{
	const uint8_t *r = rdata;
	if (byte_cnt >= 11 && byte_cnt <= 60) {
		if (dbg) {  // (whb.cpp:488-493; the verdict or the payload line follows on the same line)
			printf("#%03i %u L=%i  ", snum++, (uint32_t)time(0), byte_cnt);
			for (int n = 0; n < byte_cnt; n++)
				printf("%02x ", r[n]);
			printf(" RSSI %i ", rssi);
		}
		const int plen = r[4];
		uint32_t init;
		uint32_t want = 0, got = 0;
		bool ok = false;
		if (plen <= 60) {
			if (!whb_crc_init(r[5], &init)) {
				if (dbg >= 0)
					printf("WHB: Probably unsupported sensor type %02x! Please report\n", r[5]);
			} else {
				want = crc.calc(&rdata[4], plen - 4, init);
				got = ((uint32_t)r[plen] << 24) | (r[plen + 1] << 16) | (r[plen + 2] << 8) | r[plen + 3];
				if (want == got) {
					uint64_t id = 0;
					for (int n = 0; n < 6; n++)
						id = (id << 8) | r[5 + n];
					payload(r[5], &r[11], id, rssi);
					ok = true;
				}
			}
		}
		if (!ok) {
			bad++;
			if (dbg) {
				if (got != want)
					printf("\nWHB BAD %i RSSI %i (CRC is %08x, should be %08x, len %i, plen %i)\n", bad, rssi, got, want, byte_cnt,
					       plen);
				else
					printf("\nWHB BAD %i RSSI %i (SANITY)\n", bad, rssi);
			}
		}
	}
	sr_cnt = -1;
	sr = 0;
	byte_cnt = 0;
	synced = 0;
}
