// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Harness that drives the REAL reference hot path (compiled from /root/reference, see Makefile) and
// logs what it does, so that golden vectors can be minted and our C restatement (tfrec_oracle.c)
// can be pinned.  It contains no reference code: it includes the reference's public headers and
//   * registers the plugins in the order and with the parameters of main.cpp:171-218,
//   * restates the file-mode block loop of engine.cpp:63-93 (fread 65536 B, (u8-128)<<6,
//     downconvert::process_iq, fsk_demod::process; trailing partial block dropped),
//   * restates the -X byte replay of main.cpp:24-53 (store_bytes, flush(0), flush_storage),
//   * subclasses the reference's decoders/demodulators (their virtuals: decoder.h:39-47, 66-67)
//     to log every flush (rdata snapshot, byte_cnt, rssi, offset, sample position), every
//     store_data() record and, optionally, every store_bit().
//
// Heap memory is zero-filled (operator new below) so that the members the reference leaves
// uninitialised (last_i/last_q in all demods, whb_demod::avg_of, decoder::rdata)
// are deterministically 0 -- the same definition our restatement and the HIP path use.
//
// Modes:
//   ref_driver run  <types_hex> <thresh> <wide 0|1> <iqfile> [events_out] [dec_out] [bits 0|1]
//   ref_driver runh <types_hex> <thresh> <wide 0|1> <iqfile> <mode 0|1>   (handler "echo REC": the reference's own
//                   execute_handler() command lines, decoder.cpp:67-96, come out as "REC <args>" lines; mode = its -m;
//                   flush_storage() of every decoder at the end like main.cpp:231-234)
//   ref_driver hex  <types_hex> <hexfile> [<eventfile or ""> [<debug level: main.cpp -D / -q>]]
//   ref_driver time <types_hex> <thresh> <wide 0|1> <iqfile> <repeat>
//   ref_driver fmdev            (stdin int32[4] records -> stdout int32[2]: fm_dev, fm_dev_nrzs)
//   ref_driver iir <cutoff>     (stdin doubles -> stdout doubles through a fresh iir2)
// stdout carries the reference's own telegram lines (its printf), untouched.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <new>
#include <vector>

#include "decoder.h"
#include "dsp_stuff.h"
#include "fm_demod.h"
#include "tfa1.h"
#include "tfa2.h"
#include "whb.h"

void *operator new(size_t n) { void *p = calloc(1, n ? n : 1); if (!p) abort(); return p; }
void *operator new[](size_t n) { void *p = calloc(1, n ? n : 1); if (!p) abort(); return p; }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }

static FILE *ev_fd = NULL;
static int log_bits = 0;
static long cur_block = 0;     // block number of the block being processed (0-based)
static int cur_index = 0;      // 'index' argument of the demod() call in flight

struct spy_state {
	int slot;
	std::vector<unsigned char> bits;
};

static void log_flush(spy_state &s, const uint8_t *rdata, int byte_cnt, int rssi, int offset)
{
	if (!ev_fd)
		return;
	if (log_bits) {
		fprintf(ev_fd, "W %i %zu ", s.slot, s.bits.size());
		for (size_t n = 0; n < s.bits.size(); n++)
			fputc('0' + s.bits[n], ev_fd);
		fputc('\n', ev_fd);
		s.bits.clear();
	}
	fprintf(ev_fd, "F %i %li %i %i %i ", s.slot, cur_block * 8192 + cur_index / 2, byte_cnt, rssi, offset);
	for (int n = 0; n < 64; n++)
		fprintf(ev_fd, "%02x", rdata[n]);
	fputc('\n', ev_fd);
}

static void log_data(spy_state &s, sensordata_t &d)
{
	if (!ev_fd)
		return;
	fprintf(ev_fd, "D %i %i %llx %.17g %.17g %i %i %i %i\n", s.slot, (int)d.type,
		(unsigned long long)d.id, d.temp, d.humidity, d.sequence, d.alarm, d.rssi, d.flags);
}

#define SPY_DECODER(NAME, BASE)                                                           \
	class NAME : public BASE {                                                        \
	public:                                                                           \
		spy_state spy;                                                            \
		NAME(sensor_e t, int slot) : BASE(t) { spy.slot = slot; }                 \
		void store_bit(int bit) {                                                 \
			if (log_bits) spy.bits.push_back((unsigned char)bit);             \
			BASE::store_bit(bit);                                             \
		}                                                                         \
		void flush(int rssi, int offset = 0) {                                    \
			log_flush(spy, rdata, byte_cnt, rssi, offset);                    \
			BASE::flush(rssi, offset);                                        \
		}                                                                         \
		void store_data(sensordata_t &d) {                                        \
			log_data(spy, d);                                                 \
			BASE::store_data(d);                                              \
		}                                                                         \
	};

SPY_DECODER(spy_tfa1_decoder, tfa1_decoder)
SPY_DECODER(spy_tfa2_decoder, tfa2_decoder)
SPY_DECODER(spy_whb_decoder, whb_decoder)

class spy_tfa1_demod : public tfa1_demod {
public:
	spy_tfa1_demod(decoder *d) : tfa1_demod(d) {}
	int demod(int thresh, int pwr, int index, int16_t *iq) { cur_index = index; return tfa1_demod::demod(thresh, pwr, index, iq); }
};
class spy_tfa2_demod : public tfa2_demod {
public:
	spy_tfa2_demod(decoder *d, double spb) : tfa2_demod(d, spb) {}
	int demod(int thresh, int pwr, int index, int16_t *iq) { cur_index = index; return tfa2_demod::demod(thresh, pwr, index, iq); }
};
class spy_whb_demod : public whb_demod {
public:
	spy_whb_demod(decoder *d, double spb) : whb_demod(d, spb) {}
	int demod(int thresh, int pwr, int index, int16_t *iq) { cur_index = index; return whb_demod::demod(thresh, pwr, index, iq); }
};

// Registration: same order, sample-per-bit values and type bits as main.cpp:173-218.
static void register_demods(vector<demodulator *> &demods, int types, int debug)
{
	if (types & (1 << TFA_1)) {
		decoder *d = new spy_tfa1_decoder(TFA_1, 0);
		d->set_params(NULL, 0, debug);
		demods.push_back(new spy_tfa1_demod(d));
	}
	if (types & (1 << TFA_2)) {
		decoder *d = new spy_tfa2_decoder(TFA_2, 1);
		d->set_params(NULL, 0, debug);
		demods.push_back(new spy_tfa2_demod(d, (1536000 / 4.0) / 17240));
	}
	if (types & (1 << TFA_3)) {
		decoder *d = new spy_tfa2_decoder(TFA_3, 2);
		d->set_params(NULL, 0, debug);
		demods.push_back(new spy_tfa2_demod(d, (1536000 / 4.0) / 9600));
	}
	if (types & (1 << TX22)) {
		decoder *d = new spy_tfa2_decoder(TX22, 3);
		d->set_params(NULL, 0, debug);
		demods.push_back(new spy_tfa2_demod(d, (1536000 / 4.0) / 8842));
	}
	if (types & (1 << TFA_WHB)) {
		decoder *d = new spy_whb_decoder(TFA_WHB, 4);
		d->set_params(NULL, 0, debug);
		demods.push_back(new spy_whb_demod(d, (1536000 / 4.0) / 6000));
	}
}

#define BLOCK_BYTES 65536

static unsigned char *read_file(const char *fn, size_t *len)
{
	FILE *fd = fopen(fn, "rb");
	if (!fd) { perror(fn); exit(2); }
	fseek(fd, 0, SEEK_END);
	long sz = ftell(fd);
	fseek(fd, 0, SEEK_SET);
	unsigned char *buf = (unsigned char *)malloc(sz ? sz : 1);
	if (fread(buf, 1, sz, fd) != (size_t)sz) { perror("fread"); exit(2); }
	fclose(fd);
	*len = sz;
	return buf;
}

// engine.cpp:63-93 in file mode, on an in-memory copy of the file.
static void run_stream(const unsigned char *iq, size_t len, fsk_demod &fsk, int filter, FILE *dec_fd)
{
	downconvert dc(2);
	static int16_t data[BLOCK_BYTES];
	cur_block = 0;
	for (size_t pos = 0; pos + BLOCK_BYTES <= len; pos += BLOCK_BYTES) {
		for (int n = 0; n < BLOCK_BYTES; n++)
			data[n] = ((iq[pos + n]) - 128) << 6;
		int ld = dc.process_iq(data, BLOCK_BYTES, filter);
		if (dec_fd)
			fwrite(data, sizeof(int16_t), ld, dec_fd);
		fsk.process(data, ld);
		cur_block++;
	}
}

// The same loop for input that is already int16 (the 1.536 MS/s stream a high-rate front end hands to the
// reference's downconvert::process_iq -- BASELINE config 5): engine.cpp:85-86 without the u8 conversion of :77-78.
static void run_stream16(const int16_t *x, size_t n_values, fsk_demod &fsk, int filter, FILE *dec_fd)
{
	downconvert dc(2);
	static int16_t data[BLOCK_BYTES];
	cur_block = 0;
	for (size_t pos = 0; pos + BLOCK_BYTES <= n_values; pos += BLOCK_BYTES) {
		memcpy(data, x + pos, sizeof(data));
		int ld = dc.process_iq(data, BLOCK_BYTES, filter);
		if (dec_fd)
			fwrite(data, sizeof(int16_t), ld, dec_fd);
		fsk.process(data, ld);
		cur_block++;
	}
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;
	setvbuf(stdout, NULL, _IOFBF, 1 << 16);
	if ((!strcmp(argv[1], "run") || !strcmp(argv[1], "run16")) && argc >= 6) {
		const bool in16 = !strcmp(argv[1], "run16");  // input file: raw int16 interleaved IQ instead of u8
		int types = strtol(argv[2], NULL, 16);
		int thresh = atoi(argv[3]);
		int wide = atoi(argv[4]);
		size_t len;
		unsigned char *iq = read_file(argv[5], &len);
		if (argc > 6 && strlen(argv[6]))
			ev_fd = fopen(argv[6], "w");
		FILE *dec_fd = NULL;
		if (argc > 7 && strlen(argv[7]))
			dec_fd = fopen(argv[7], "wb");
		if (argc > 8)
			log_bits = atoi(argv[8]);
		vector<demodulator *> demods;
		register_demods(demods, types, 0);
		fsk_demod fsk(&demods, thresh, 0);
		puts("---");  // separates constructor chatter from telegram lines
		if (in16)
			run_stream16((const int16_t *)iq, len / 2, fsk, wide, dec_fd);
		else
			run_stream(iq, len, fsk, wide, dec_fd);
		if (ev_fd) fclose(ev_fd);
		if (dec_fd) fclose(dec_fd);
		fflush(stdout);
		return 0;
	}
	if (!strcmp(argv[1], "runh") && argc >= 7) {
		int types = strtol(argv[2], NULL, 16);
		int thresh = atoi(argv[3]);
		int wide = atoi(argv[4]);
		int mode = atoi(argv[6]);
		size_t len;
		unsigned char *iq = read_file(argv[5], &len);
		vector<demodulator *> demods;
		register_demods(demods, types, -1);  // quiet: only the handler's lines
		static char handler[] = "echo REC";
		if (argc > 7)
			ev_fd = fopen(argv[7], "w");
		for (size_t n = 0; n < demods.size(); n++)
			demods[n]->dec->set_params(handler, mode, -1);
		fsk_demod fsk(&demods, thresh, -1);
		puts("---");
		fflush(stdout);
		run_stream(iq, len, fsk, wide, NULL);
		fflush(stdout);
		for (size_t n = 0; n < demods.size(); n++)
			demods[n]->dec->flush_storage();
		fflush(stdout);
		return 0;
	}
	if (!strcmp(argv[1], "hex") && argc >= 4) {
		int types = strtol(argv[2], NULL, 16);
		vector<demodulator *> demods;
		register_demods(demods, types, argc > 5 ? atoi(argv[5]) : 0);  // [5]: the decoders' debug level (main.cpp -D)
		puts("---");
		FILE *fd = fopen(argv[3], "r");
		if (!fd) { perror(argv[3]); return 2; }
		if (argc > 4 && strlen(argv[4]))
			ev_fd = fopen(argv[4], "w");
		char buf[1024];
		while (fgets(buf, sizeof(buf), fd)) {
			if (buf[0] == '#')
				continue;
			unsigned char dbuf[512];
			unsigned int len = 0;
			char *dp = buf, *x;
			while ((x = strsep(&dp, " ")) && len < sizeof(dbuf))
				if (*x != 0 && *x != '\n')
					dbuf[len++] = strtol(x, NULL, 16);
			for (size_t n = 0; n < demods.size(); n++) {
				demods[n]->dec->store_bytes(dbuf, len);
				demods[n]->dec->flush(0);
				puts("");
				demods[n]->dec->flush_storage();
			}
		}
		fclose(fd);
		if (ev_fd) fclose(ev_fd);
		fflush(stdout);
		return 0;
	}
	if (!strcmp(argv[1], "time") && argc >= 7) {
		int types = strtol(argv[2], NULL, 16);
		int thresh = atoi(argv[3]);
		int wide = atoi(argv[4]);
		size_t len;
		unsigned char *iq = read_file(argv[5], &len);
		int repeat = atoi(argv[6]);
		FILE *sink = freopen("/dev/null", "w", stdout);
		(void)sink;
		struct timespec t0, t1;
		double total = 0;
		size_t samples = 0;
		for (int r = 0; r < repeat; r++) {
			vector<demodulator *> demods;
			register_demods(demods, types, -1);
			fsk_demod fsk(&demods, thresh, -1);
			clock_gettime(CLOCK_MONOTONIC, &t0);
			run_stream(iq, len, fsk, wide, NULL);
			clock_gettime(CLOCK_MONOTONIC, &t1);
			total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
			samples += (len / BLOCK_BYTES) * (BLOCK_BYTES / 2);
		}
		fprintf(stderr, "{\"seconds\": %.6f, \"samples\": %zu, \"msps\": %.3f}\n", total, samples,
			samples / total / 1e6);
		return 0;
	}
	// Unit probes of the reference's public DSP functions (dsp_stuff.h:20-31, 54-55), binary stdin->stdout.
	if (!strcmp(argv[1], "fmdev")) { // in: int32[4] per record; out: int32 fm_dev, int32 fm_dev_nrzs
		int32_t q[4];
		while (fread(q, sizeof(q), 1, stdin) == 1) {
			int32_t r[2] = { fm_dev(q[0], q[1], q[2], q[3]), fm_dev_nrzs(q[0], q[1], q[2], q[3]) };
			fwrite(r, sizeof(r), 1, stdout);
		}
		fflush(stdout);
		return 0;
	}
	if (!strcmp(argv[1], "iir") && argc >= 3) { // in: double per record; out: double iir2::step
		iir2 f(atof(argv[2]));
		double v;
		while (fread(&v, sizeof(v), 1, stdin) == 1) {
			double y = f.step(v);
			fwrite(&y, sizeof(y), 1, stdout);
		}
		fflush(stdout);
		return 0;
	}
	fprintf(stderr, "usage: see header comment\n");
	return 1;
}
