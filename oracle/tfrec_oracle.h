/* oracle/tfrec_oracle.h -- TEST INFRASTRUCTURE ONLY (see tfrec_oracle.c header). */
#ifndef TFREC_ORACLE_H
#define TFREC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSLOTS 5 /* TFA_1, TFA_2, TFA_3, TX22, WHB: registration order of main.cpp:173-218 */
#define ORC_BLOCK_BYTES 65536
#define ORC_BLOCK_DEC 8192 /* decimated IQ pairs per block */

typedef struct {
	int16_t slot;       /* 0..4 */
	int16_t status;     /* what the decoder's flush() did with it: 0 = shorter than a telegram (tfa1.cpp:49, tfa2.cpp:76/222,
			       whb.cpp:484), 1 = passed its CRC + sanity tests (tfa1.cpp:63-73, tfa2.cpp:93/237, whb.cpp:506-510),
			       2 = rejected by them */
	int32_t byte_cnt;   /* decoder byte_cnt when flush() was entered */
	int32_t rssi_db;    /* first flush() argument as the demodulator computed it */
	int32_t offset;     /* second flush() argument */
	int64_t end_sample; /* decimated sample index (since stream start) at which flush fired */
	int64_t rssi_raw;   /* demodulator's raw rssi accumulator (tfa1/tfa2: int, whb: exact integer sum) */
	uint8_t rdata[64];  /* decoder rdata[0..64) when flush() was entered */
} orc_event_t;

typedef struct {
	int32_t slot;
	int32_t type; /* sensor_e value, decoder.h:11-19 */
	uint64_t id;
	double temp;
	double humidity;
	int32_t sequence;
	int32_t alarm;
	int32_t rssi;
	int32_t flags;
} orc_data_t;

typedef struct orc orc_t;

/* types_mask: bit n = sensor_e n (TFA_1=0, TFA_2=1, TFA_3=2, TX22=3, TFA_WHB=5), as main.cpp -T.
 * thresh: 0 = auto (fm_demod.cpp:23-27).  wide: -W filter (dsp_stuff.cpp:176-178). */
orc_t *orc_create(int types_mask, int thresh, int wide);
void orc_destroy(orc_t *o);
void orc_set_log_bits(orc_t *o, int on);
void orc_set_keep_dec(orc_t *o, int on);
void orc_set_quiet(orc_t *o, int on); /* timing runs: no text, no event/data/bit/dec logs */

/* Feed whole 65536-byte blocks (engine.cpp:67-86); a trailing partial block is dropped like the
 * reference does.  Returns the number of blocks consumed. */
long orc_process(orc_t *o, const uint8_t *iq, size_t nbytes);

/* The same for blocks of 65536 int16 values (input that is already (I,Q) int16 at 1.536 MS/s). */
long orc_process_s16(orc_t *o, const int16_t *x, size_t n_values);
/* BASELINE config 5 front end (defined here, no reference counterpart): 15.36 MS/s u8 IQ -> 1.536 MS/s int16 IQ */
extern const int16_t orc_taps10[60];
void orc_decim10(const uint8_t *iq, size_t n_in_complex, int16_t *out);

/* CPU-baseline helper: n_jobs quiet receivers (job k reads stream k % n_streams) over `threads` OpenMP threads; seconds */
double orc_time_many(int types_mask, int thresh, int wide, const uint8_t *iq, size_t stride, size_t nbytes, int n_streams,
		     int n_jobs, int threads);

/* Batched checker: n_streams fresh receivers over `threads` OpenMP threads; events of stream s -> out[s*cap ..] (first
 * cap), counts[s] = events the stream produced */
void orc_process_many(int types_mask, int thresh, int wide, const uint8_t *iq, size_t stride, size_t nbytes, int n_streams,
		      int threads, orc_event_t *out, size_t cap, int64_t *counts);

/* The same for streams that continue across several input arrays: part p (bases[p] + s * strides[p], nbytes[p]) is run
 * reps[p] times in a row on the stream's one receiver; events from part keep_from on are returned */
void orc_process_parts(int types_mask, int thresh, int wide, int n_parts, const uint8_t *const *bases, const size_t *strides,
		       const size_t *nbytes, const int *reps, int keep_from, int n_streams, int threads, orc_event_t *out,
		       size_t cap, int64_t *counts);

/* -X replay (main.cpp:24-53): store_bytes + flush(0) on every registered decoder. */
void orc_hex(orc_t *o, const uint8_t *bytes, int len);

size_t orc_num_events(const orc_t *o);
const orc_event_t *orc_events(const orc_t *o);
size_t orc_num_data(const orc_t *o);
const orc_data_t *orc_data(const orc_t *o);
const char *orc_text(const orc_t *o); /* telegram lines the reference prints at dbg=0 */
size_t orc_text_len(const orc_t *o);
size_t orc_num_dec(const orc_t *o); /* int16 count of the kept decimated stream */
const int16_t *orc_dec(const orc_t *o);
/* bit log: per flush "slot nbits bits..." records, as text */
const char *orc_bits_text(const orc_t *o);
int orc_thresh(const orc_t *o); /* current trigger threshold (auto mode moves it) */
uint64_t orc_atan_uncertain(const orc_t *o); /* samples whose fm_dev truncation is within 1e-6 of flipping */
void orc_clear_logs(orc_t *o);

/* unit-level entry points (pinned against the real reference functions by ref_driver probes) */
int orc_fm_dev(int ar, int aj, int br, int bj);
int orc_fm_dev_nrzs(int ar, int aj, int br, int bj);
uint8_t orc_crc8(const uint8_t *d, int len);
uint32_t orc_crc32(const uint8_t *d, int len, uint32_t init);
void orc_iir_coeffs(double cutoff, double out[5]); /* b0 b1 b2 a1 a2 */
/* run a fresh biquad over n inputs */
void orc_iir_run(double cutoff, const double *in, double *out, size_t n);
/* stateless decimator over a whole stream from zero history: in u8 interleaved IQ (n complex samples,
 * n % 4 == 0), out interleaved int16 IQ (n/4 pairs). */
void orc_decimate(const uint8_t *iq, size_t n_complex, int wide, int16_t *out);

#ifdef __cplusplus
}
#endif
#endif
