/* tfrec_amd/synth/iqgen.c -- deterministic synthetic 8-bit IQ streams for tests and bench.py.
 *
 * Signal source only: it produces the u8 interleaved IQ bytes an RTL-SDR would deliver at 1.536 MS/s
 * (sdr.cpp:233-240 writes exactly this format with -S; engine.cpp:67-81 reads it with -L).  It follows
 * the recipes of SURVEY.md Appendix C / section 8(d): silence, then bursts round-robin over the five
 * protocols with random valid payloads, amplitude U(30,110) LSB, carrier offset U(-10,10) kHz, random
 * start phase, inter-burst silence U(20000,60000) samples, additive noise, clipping to [0,255].
 *
 * Everything is INTEGER arithmetic (fixed-point NCO with a polynomial sine, box-cascade pulse
 * shaping, Irwin-Hall noise from a 64-bit xorshift generator), so the bytes are identical on every
 * machine and compiler -- golden vectors minted from them stay valid on the GPU box.
 *
 * Frame layouts are the on-air formats documented in the reference's protocol notes
 * (tfa1.cpp:6-31, tfa2.cpp:6-52, whb.cpp:9-46) and SURVEY.md Appendix C.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FS_BASE 1536000
#define BLOCK_SAMPLES 32768

typedef struct {
	int32_t proto;      /* 0 TFA_1, 1 TFA_2, 2 TFA_3, 3 TX22, 4 WHB */
	int32_t nbytes;     /* frame length */
	int64_t start;      /* first input sample of the burst */
	int64_t length;     /* burst length in input samples */
	int32_t amp_q4;     /* amplitude in 1/16 LSB */
	int32_t f0_hz;      /* carrier offset */
	uint8_t frame[64];  /* frame bytes incl. sync and CRC */
} iqgen_truth_t;

/* ---- PRNG */
typedef struct { uint64_t s; } rng_t;
static uint64_t rng_next(rng_t *r)
{
	uint64_t x = r->s;
	x ^= x >> 12;
	x ^= x << 25;
	x ^= x >> 27;
	r->s = x;
	return x * 0x2545F4914F6CDD1DULL;
}
static void rng_seed(rng_t *r, uint64_t seed, uint64_t stream)
{
	uint64_t z = seed * 0x9E3779B97F4A7C15ULL + stream * 0xBF58476D1CE4E5B9ULL + 0x94D049BB133111EBULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	r->s = z ? z : 0x1234567887654321ULL;
	for (int i = 0; i < 4; i++)
		rng_next(r);
}
static uint32_t rng_range(rng_t *r, uint32_t lo, uint32_t hi) /* inclusive */
{
	return lo + (uint32_t)((rng_next(r) >> 32) % (uint64_t)(hi - lo + 1));
}

/* ---- fixed-point sine: phase 32 bit = one turn, result Q15 */
static int32_t sin_q15(uint32_t ph)
{
	/* quadrant folding, x in [0, 2^30] = quarter turn, then odd polynomial of sin(pi/2 * x) */
	uint32_t q = ph >> 30;
	uint32_t f = ph & 0x3fffffffu;
	int64_t x = (q & 1) ? (int64_t)(0x40000000u - f) : (int64_t)f; /* Q30 in [0,1] */
	int64_t x2 = (x * x) >> 30;
	/* coefficients of sin(pi/2 x) ~ c1 x - c3 x^3 + c5 x^5 - c7 x^7, Q30 */
	const int64_t c1 = 1686629713, c3 = 693598342, c5 = 85568996, c7 = 5026909;
	int64_t p = c5 - ((c7 * x2) >> 30);
	p = c3 - ((p * x2) >> 30);
	p = c1 - ((p * x2) >> 30);
	int64_t s = (p * x) >> 30; /* Q30 */
	int32_t v = (int32_t)((s + (1 << 14)) >> 15);
	if (v > 32767)
		v = 32767;
	return (q & 2) ? -v : v;
}

/* ---- CRCs of the on-air formats (poly 0x31 MSB-first init 0; poly 0x04c11db7 MSB-first, init per type) */
static uint8_t crc8_31(const uint8_t *d, int n)
{
	uint8_t c = 0;
	for (int i = 0; i < n; i++) {
		c ^= d[i];
		for (int b = 0; b < 8; b++)
			c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x31) : (uint8_t)(c << 1);
	}
	return c;
}
static uint32_t crc32_04c11db7(const uint8_t *d, int n, uint32_t c)
{
	for (int i = 0; i < n; i++) {
		c ^= (uint32_t)d[i] << 24;
		for (int b = 0; b < 8; b++)
			c = (c & 0x80000000u) ? ((c << 1) ^ 0x04c11db7u) : (c << 1);
	}
	return c;
}

/* ---- frame builders */
static int bcd_temp(rng_t *r, uint8_t *hi_nibble, uint8_t *lo_byte)
{
	int t = (int)rng_range(r, 0, 999); /* (degC+40)*10 */
	*hi_nibble = (uint8_t)(t / 100);
	*lo_byte = (uint8_t)((((t / 10) % 10) << 4) | (t % 10));
	return t;
}

static int frame_tfa1(rng_t *r, uint8_t *f)
{
	uint8_t th, tl;
	bcd_temp(r, &th, &tl);
	uint32_t id = rng_range(r, 1, 0x7fff);
	f[0] = 0x2d; f[1] = 0xd4;
	f[2] = (uint8_t)(id >> 8); f[3] = (uint8_t)id;
	f[4] = 0x80 | th; f[5] = tl;
	f[6] = (uint8_t)rng_range(r, 1, 99);
	f[7] = 0x60 | (rng_range(r, 0, 7) == 0 ? 0x80 : 0);
	f[8] = (uint8_t)(rng_range(r, 0, 15) << 4);
	f[9] = 0x56;
	f[10] = crc8_31(f + 2, 8);
	return 11;
}

static int frame_tfa23(rng_t *r, uint8_t *f)
{
	uint8_t th, tl;
	bcd_temp(r, &th, &tl);
	uint32_t id = rng_range(r, 0, 63) << 2;
	f[0] = 0x2d; f[1] = 0xd4;
	f[2] = 0x90 | (uint8_t)(id >> 4);
	f[3] = (uint8_t)((id & 0xf) << 4) | th;
	f[4] = tl;
	f[5] = (uint8_t)rng_range(r, 1, 99);
	f[6] = crc8_31(f + 2, 4);
	return 7;
}

static int frame_tx22(rng_t *r, uint8_t *f)
{
	uint32_t id = rng_range(r, 0, 63);
	int num = (int)rng_range(r, 1, 5);
	f[0] = 0x2d; f[1] = 0xd4;
	f[2] = 0xa0 | (uint8_t)(id >> 2);
	f[3] = (uint8_t)((id & 3) << 6) | 0x10 | (rng_range(r, 0, 7) == 0 ? 0x08 : 0) | (uint8_t)num;
	for (int n = 0; n < num; n++) {
		uint8_t *w = f + 4 + 2 * n;
		int v;
		switch (n) {
		case 0: { uint8_t th, tl; bcd_temp(r, &th, &tl); w[0] = th; w[1] = tl; break; }
		case 1: v = (int)rng_range(r, 1, 99); w[0] = 0x10; w[1] = (uint8_t)(((v / 10) << 4) | (v % 10)); break;
		case 2: v = (int)rng_range(r, 0, 4095); w[0] = 0x20 | (uint8_t)(v >> 8); w[1] = (uint8_t)v; break;
		case 3: w[0] = 0x30 | (uint8_t)rng_range(r, 0, 15); w[1] = (uint8_t)rng_range(r, 0, 255); break;
		default: v = (int)rng_range(r, 0, 4095); w[0] = 0x40 | (uint8_t)(v >> 8); w[1] = (uint8_t)v; break;
		}
	}
	f[4 + 2 * num] = crc8_31(f + 2, 2 + 2 * num);
	return 5 + 2 * num;
}

static const struct { uint8_t type; uint8_t paylen; uint32_t init; } whb_types[11] = {
	{ 0x02, 6, 0x97d97a26 },  { 0x03, 11, 0xf59c5a1e }, { 0x04, 12, 0x98e1d11f }, { 0x06, 14, 0xa7a41254 },
	{ 0x07, 18, 0x3303fb1d }, { 0x08, 26, 0x29f0f49b }, { 0x09, 14, 0xa7a41254 }, { 0x0b, 27, 0xe7720ae4 },
	{ 0x10, 10, 0x62d0afc1 }, { 0x11, 34, 0x8cba0708 }, { 0x12, 9, 0x5a9e30ae },
};

static int frame_whb(rng_t *r, uint8_t *f)
{
	int k = (int)rng_range(r, 0, 10);
	int pl = whb_types[k].paylen;
	f[0] = 0x4b; f[1] = 0x2d; f[2] = 0xd4; f[3] = 0x2b;
	f[4] = (uint8_t)(pl + 11);
	f[5] = whb_types[k].type;
	for (int n = 0; n < 5; n++)
		f[6 + n] = (uint8_t)rng_range(r, 0, 255);
	for (int n = 0; n < pl; n++)
		f[11 + n] = (uint8_t)rng_range(r, 0, 255);
	uint32_t c = crc32_04c11db7(f + 4, pl + 7, whb_types[k].init);
	f[11 + pl] = (uint8_t)(c >> 24);
	f[12 + pl] = (uint8_t)(c >> 16);
	f[13 + pl] = (uint8_t)(c >> 8);
	f[14 + pl] = (uint8_t)c;
	return pl + 15;
}

/* ---- burst synthesis */

typedef struct {
	int32_t *lvl;  /* per-sample level / frequency control, Q12 */
	int32_t *tmp;
	size_t cap;
} work_t;

static void work_reserve(work_t *w, size_t n)
{
	if (n <= w->cap)
		return;
	w->cap = n + 4096;
	w->lvl = (int32_t *)realloc(w->lvl, w->cap * sizeof(int32_t));
	w->tmp = (int32_t *)realloc(w->tmp, w->cap * sizeof(int32_t));
}

/* centred boxcar of width wd over a[0..n), edges padded with the edge value; integer division */
static void boxcar(const int32_t *a, int32_t *o, size_t n, int wd)
{
	if (wd <= 1) {
		memcpy(o, a, n * sizeof(int32_t));
		return;
	}
	int h0 = wd / 2, h1 = wd - 1 - h0;
	int64_t acc = 0;
	for (int k = -h0; k <= h1; k++) {
		long idx = k < 0 ? 0 : (k >= (long)n ? (long)n - 1 : k);
		acc += a[idx];
	}
	for (size_t i = 0; i < n; i++) {
		o[i] = (int32_t)((acc >= 0 ? acc + wd / 2 : acc - wd / 2) / wd);
		long out = (long)i - h0, in = (long)i + h1 + 1;
		acc -= a[out < 0 ? 0 : out];
		acc += a[in >= (long)n ? (long)n - 1 : in];
	}
}

static void smooth3(work_t *w, size_t n, int wd)
{
	boxcar(w->lvl, w->tmp, n, wd);
	boxcar(w->tmp, w->lvl, n, wd);
	boxcar(w->lvl, w->tmp, n, wd);
	memcpy(w->lvl, w->tmp, n * sizeof(int32_t));
}

static inline void put_bits_lsb(uint8_t *bits, size_t *nb, const uint8_t *f, int nbytes)
{
	for (int i = 0; i < nbytes; i++)
		for (int b = 0; b < 8; b++)
			bits[(*nb)++] = (f[i] >> b) & 1;
}
static inline void put_bits_msb(uint8_t *bits, size_t *nb, const uint8_t *f, int nbytes)
{
	for (int i = 0; i < nbytes; i++)
		for (int b = 7; b >= 0; b--)
			bits[(*nb)++] = (f[i] >> b) & 1;
}

/* add signal value (Q16 LSB) into the accumulation arrays */
typedef struct {
	int32_t *si, *sq; /* Q16 signal per rail for the whole stream chunk being built (sparse: only bursts) */
} sig_t;

static const int baud_tab[5] = { 38400, 17240, 9600, 8842, 6000 };

/* returns burst length in samples; writes Q16 signal into si/sq[start..start+len) */
/* Planted faults (tests of the decoders' acceptance verdicts through the whole path; no random numbers are drawn, so the
 * signal of every other burst stays what it was): 1 = the checksum byte(s) do not match, 2 = checksum right but a field
 * the decoder's sanity test looks at is wrong (TFA_1: tfa1.cpp:66-73 wants r[9] == 0x56; TX22: tfa2.cpp:76 wants the 0xa
 * nibble; WHB: a type whb.cpp:50-62 does not know; TFA_2/3 have no such test: a payload byte changed under a stale CRC),
 * 3 = the frame cut short by three bytes */
static int corrupt_frame(int proto, uint8_t *f, int nbytes, int mode)
{
	if (mode == 1) {
		f[nbytes - 1] ^= 0x5a;
	} else if (mode == 2) {
		switch (proto) {
		case 0: f[9] = 0x55; f[10] = crc8_31(f + 2, 8); break;
		case 1:
		case 2: f[4] ^= 0x10; break;
		case 3: f[2] = (uint8_t)(0xb0 | (f[2] & 0x0f)); f[nbytes - 1] = crc8_31(f + 2, nbytes - 3); break;
		default: f[5] = 0x05; break;
		}
	} else if (mode == 3 && nbytes > 6) {
		nbytes -= 3;
	}
	return nbytes;
}

static int64_t synth_burst(rng_t *r, work_t *w, int proto, int64_t start, int64_t total, int32_t *si, int32_t *sq,
			   iqgen_truth_t *tr, int64_t fs, int corrupt)
{
	const int64_t FS = fs;
	uint8_t frame[64];
	uint8_t bits[1024];
	size_t nb = 0;
	int nbytes;
	memset(frame, 0, sizeof(frame));
	switch (proto) {
	case 0:
		nbytes = corrupt_frame(proto, frame, frame_tfa1(r, frame), corrupt);
		for (int i = 0; i < 200; i++) bits[nb++] = 0;
		put_bits_lsb(bits, &nb, frame, nbytes);
		for (int i = 0; i < 48; i++) bits[nb++] = 0;
		break;
	case 1:
	case 2:
	case 3: {
		nbytes = corrupt_frame(proto, frame, proto == 3 ? frame_tx22(r, frame) : frame_tfa23(r, frame), corrupt);
		int pre = proto == 1 ? 4 : (proto == 2 ? 12 : 8);
		for (int i = 0; i < pre; i++) { bits[nb++] = 1; bits[nb++] = 0; }
		put_bits_msb(bits, &nb, frame, nbytes);
		bits[nb++] = 0; bits[nb++] = 1; /* short tail so the last data edge exists */
		break;
	}
	default: {
		nbytes = corrupt_frame(proto, frame, frame_whb(r, frame), corrupt);
		uint8_t d[1024];
		size_t nd = 0;
		for (int i = 0; i < 200; i++) d[nd++] = 1;
		put_bits_lsb(d, &nd, frame, nbytes);
		for (int i = 0; i < 24; i++) d[nd++] = 1;
		/* G3RUH scrambler s[t] = d[t]^s[t-12]^s[t-17] (inverse of whb.cpp:579-580) */
		for (size_t t = 0; t < nd; t++) {
			uint8_t s = d[t];
			if (t >= 12) s ^= bits[t - 12];
			if (t >= 17) s ^= bits[t - 17];
			bits[t] = s;
		}
		nb = nd;
		break;
	}
	}
	int baud = baud_tab[proto];
	int64_t len = ((int64_t)nb * FS + baud - 1) / baud;
	if (start + len > total)
		return -1;
	int32_t amp_q4 = (int32_t)rng_range(r, 30 * 16, 110 * 16);
	int32_t f0 = (int32_t)rng_range(r, 0, 20000) - 10000;
	uint32_t ph = (uint32_t)rng_next(r);
	int32_t fdev = (int32_t)rng_range(r, 30000, 60000);
	work_reserve(w, (size_t)len);
	int psk = (proto == 0 || proto == 4);
	if (psk) {
		/* NRZS: level toggles on every 0 bit; shaped +-1 envelope times carrier */
		int lv = 1;
		int64_t n = 0;
		for (size_t b = 0; b < nb; b++) {
			if (!bits[b]) lv = -lv;
			int64_t end = ((int64_t)(b + 1) * FS) / baud;
			for (; n < end && n < len; n++) w->lvl[n] = lv * 4096;
		}
		for (; n < len; n++) w->lvl[n] = lv * 4096;
		int spb = FS / baud;
		smooth3(w, (size_t)len, (spb * 17 + 32) / 64); /* box cascade ~ Gaussian BT 1.0 */
		uint32_t dph = (uint32_t)(int32_t)(((int64_t)f0 << 32) / FS);
		for (int64_t k = 0; k < len; k++) {
			int64_t a = (int64_t)amp_q4 * w->lvl[k]; /* Q4 * Q12 = Q16 */
			si[start + k] = (int32_t)((a * sin_q15(ph + 0x40000000u)) >> 15);
			sq[start + k] = (int32_t)((a * sin_q15(ph)) >> 15);
			ph += dph;
		}
	} else {
		/* CPFSK: instantaneous frequency f0 +- fdev, lightly smoothed (BT ~0.5) */
		int64_t n = 0;
		for (size_t b = 0; b < nb; b++) {
			int64_t end = ((int64_t)(b + 1) * FS) / baud;
			for (; n < end && n < len; n++) w->lvl[n] = bits[b] ? 4096 : -4096;
		}
		for (; n < len; n++) w->lvl[n] = 0;
		int spb = FS / baud;
		smooth3(w, (size_t)len, (spb * 34 + 32) / 64);
		int64_t a = (int64_t)amp_q4 << 12; /* Q16 */
		int64_t w0 = ((int64_t)f0 << 32) / FS, wd = ((int64_t)fdev << 32) / FS;
		for (int64_t k = 0; k < len; k++) {
			si[start + k] = (int32_t)((a * sin_q15(ph + 0x40000000u)) >> 15);
			sq[start + k] = (int32_t)((a * sin_q15(ph)) >> 15);
			ph += (uint32_t)(int32_t)(w0 + ((wd * w->lvl[k]) >> 12));
		}
	}
	if (tr) {
		tr->proto = proto;
		tr->nbytes = nbytes;
		tr->start = start;
		tr->length = len;
		tr->amp_q4 = amp_q4;
		tr->f0_hz = f0;
		memcpy(tr->frame, frame, 64);
	}
	return len;
}

/* noise + quantisation of [n0,n1): si/sq may be NULL (pure silence) */
static void quantise(rng_t *r, const int32_t *si, const int32_t *sq, int64_t n, int noise_q8, uint8_t *out)
{
	/* Irwin-Hall(4) over bytes: mean 510, sigma sqrt(4*(256^2-1)/12) = 147.80; scale to Q16 LSB */
	const int64_t k = ((int64_t)noise_q8 * 65536 * 1000) / (256 * 147802LL);
	for (int64_t i = 0; i < n; i++) {
		uint64_t u = rng_next(r);
		int32_t n1 = (int32_t)((u & 0xff) + ((u >> 8) & 0xff) + ((u >> 16) & 0xff) + ((u >> 24) & 0xff)) - 510;
		int32_t n2 = (int32_t)(((u >> 32) & 0xff) + ((u >> 40) & 0xff) + ((u >> 48) & 0xff) + ((u >> 56) & 0xff)) - 510;
		int64_t vi = (si ? si[i] : 0) + n1 * k + (128LL << 16) + 32768;
		int64_t vq = (sq ? sq[i] : 0) + n2 * k + (128LL << 16) + 32768;
		vi >>= 16;
		vq >>= 16;
		out[2 * i] = (uint8_t)(vi < 0 ? 0 : (vi > 255 ? 255 : vi));
		out[2 * i + 1] = (uint8_t)(vq < 0 ? 0 : (vq > 255 ? 255 : vq));
	}
}

/* Generate one stream of n_blocks*65536*rate_mult bytes at rate_mult * 1.536 MS/s (rate_mult 1: what an RTL-SDR
 * delivers; 10: the 15.36 MS/s input of BASELINE config 5 -- same recipe, every duration in samples scaled).
 * proto_mask: bit p enables protocol p (0..4).  noise_q8: noise sigma in 1/256 LSB (256 = 1.0 LSB).  Returns the
 * number of planted bursts. */
int iqgen_stream_ex(uint64_t seed, uint32_t stream, int n_blocks, int proto_mask, int noise_q8, uint8_t *out,
		    iqgen_truth_t *truth, int truth_cap, int rate_mult, int corrupt_every);

int iqgen_stream_rate(uint64_t seed, uint32_t stream, int n_blocks, int proto_mask, int noise_q8, uint8_t *out,
		      iqgen_truth_t *truth, int truth_cap, int rate_mult)
{
	return iqgen_stream_ex(seed, stream, n_blocks, proto_mask, noise_q8, out, truth, truth_cap, rate_mult, 0);
}

/* corrupt_every = c > 0: bursts 0, c, 2c, ... carry a planted fault (corrupt_frame modes 1, 2, 3 in turn) */
int iqgen_stream_ex(uint64_t seed, uint32_t stream, int n_blocks, int proto_mask, int noise_q8, uint8_t *out,
		    iqgen_truth_t *truth, int truth_cap, int rate_mult, int corrupt_every)
{
	rng_t r, rn;
	rng_seed(&r, seed, stream);
	rng_seed(&rn, seed ^ 0x5851F42D4C957F2DULL, stream);
	const int64_t fs = (int64_t)FS_BASE * rate_mult;
	const int64_t block_samples = (int64_t)BLOCK_SAMPLES * rate_mult;
	int64_t total = (int64_t)n_blocks * block_samples;
	int32_t *si = (int32_t *)calloc((size_t)total, sizeof(int32_t));
	int32_t *sq = (int32_t *)calloc((size_t)total, sizeof(int32_t));
	work_t w = { 0, 0, 0 };
	int nt = 0;
	int64_t pos = 40000 * (int64_t)rate_mult;
	int proto = (int)(stream % 5);
	int64_t limit = total - block_samples - 4096 * (int64_t)rate_mult; /* keep the last block silent so windows time out */
	if (proto_mask & 0x1f) {
		while (pos < limit) {
			while (!(proto_mask & (1 << proto)))
				proto = (proto + 1) % 5;
			iqgen_truth_t tr;
			const int fault = (corrupt_every > 0 && nt % corrupt_every == 0) ? 1 + (nt / corrupt_every) % 3 : 0;
			int64_t len = synth_burst(&r, &w, proto, pos, limit, si, sq, &tr, fs, fault);
			if (len < 0)
				break;
			if (truth && nt < truth_cap)
				truth[nt] = tr;
			nt++;
			pos += len + (int64_t)rng_range(&r, 20000, 60000) * rate_mult;
			proto = (proto + 1) % 5;
		}
	}
	quantise(&rn, si, sq, total, noise_q8, out);
	free(si);
	free(sq);
	free(w.lvl);
	free(w.tmp);
	return nt;
}

int iqgen_stream(uint64_t seed, uint32_t stream, int n_blocks, int proto_mask, int noise_q8, uint8_t *out,
		 iqgen_truth_t *truth, int truth_cap)
{
	return iqgen_stream_rate(seed, stream, n_blocks, proto_mask, noise_q8, out, truth, truth_cap, 1);
}

/* Batch: streams first_stream .. first_stream+n_streams-1, contiguous in out. OpenMP over streams. */
int iqgen_batch(uint64_t seed, uint32_t first_stream, int n_streams, int n_blocks, int proto_mask, int noise_q8,
		uint8_t *out)
{
	long total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
	for (int s = 0; s < n_streams; s++)
		total += iqgen_stream(seed, first_stream + (uint32_t)s, n_blocks, proto_mask, noise_q8,
				      out + (size_t)s * n_blocks * 65536, NULL, 0);
	return (int)total;
}
