/* include/tfrec_amd.h -- C ABI of the MI355X-native IQ->telegram hot path.
 *
 * This is the drop-in boundary for baycom/tfrec's inner loop.  In the reference one receiver does, per
 * 65536-byte block of raw 8-bit IQ (engine.cpp:63-93):
 *
 *     datab[n] = (buf[n]-128)<<6;                      engine.cpp:77-78
 *     ld = dc.process_iq(data, len, filter_type);      engine.cpp:85   (dsp_stuff.cpp:243-264)
 *     fsk->process(data, ld);                          engine.cpp:86   (fm_demod.cpp:34-74)
 *         -> demodulator::start / ::demod              decoder.h:65-67 (tfa1.cpp:143, tfa2.cpp:346, whb.cpp:632)
 *         -> decoder::store_bit ... decoder::flush     decoder.h:39-40 (tfa1.cpp:120/47, tfa2.cpp:281/64, whb.cpp:566/477)
 *
 * Here the same work is done for a BATCH of independent streams on one GPU: tfrec_amd_submit_*()
 * replaces the three calls above for every stream of the batch, and tfrec_amd_drain_events() hands
 * back, per (stream, demodulator slot) and in order, what each reference decoder would have held at
 * the moment demodulator::demod() called decoder::flush(rssi, offset): byte_cnt, rdata[], the raw
 * RSSI accumulator and the frequency offset.  A host adapter replays each event into an (unchanged)
 * reference decoder object with decoder::store_bytes(ev.rdata, ev.byte_cnt) + decoder::flush(
 * tfrec_amd_rssi_db(...), ev.offset) -- the reference's own "-X" test entry (main.cpp:45-49,
 * decoder.cpp:35-40) -- see INTEGRATION.md.
 *
 * Plain C types only; caller-owned buffers; int return codes (0 = ok, <0 = TFREC_AMD_E_*); no
 * exceptions cross this boundary.  One host thread per context; contexts are independent (one per GPU
 * in a multi-GPU job: streams shard by index, no collective is involved).
 */
#ifndef TFREC_AMD_H
#define TFREC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFREC_AMD_BLOCK_BYTES 65536 /* RLS, engine.cpp:68 */
#define TFREC_AMD_FIFO_DEPTH 4  /* submits that may wait to be drained (tfrec_amd_drain_events); a property of the built
				   library: tfrec_amd_fifo_depth() reports the value it was compiled with */
#define TFREC_AMD_BLOCK_BYTES_10X 655360 /* one block of a 15.36 MS/s stream (TFREC_AMD_F_INPUT_10X) */
#define TFREC_AMD_BLOCK_DEC 8192    /* decimated IQ pairs per block (4:1, dsp_stuff.cpp:243-264) */
#define TFREC_AMD_NSLOTS 5

/* demodulator slots = registration order of main.cpp:173-218 */
enum { TFREC_AMD_SLOT_TFA1 = 0, TFREC_AMD_SLOT_TFA2 = 1, TFREC_AMD_SLOT_TFA3 = 2, TFREC_AMD_SLOT_TX22 = 3,
       TFREC_AMD_SLOT_WHB = 4 };

/* error codes */
enum {
	TFREC_AMD_OK = 0,
	TFREC_AMD_E_INVAL = -1,     /* bad argument / unsupported configuration */
	TFREC_AMD_E_NOMEM = -2,     /* host or device allocation failed */
	TFREC_AMD_E_HIP = -3,       /* a HIP runtime call failed (no device, launch failure, ...) */
	TFREC_AMD_E_OVERFLOW = -4,  /* event buffer too small: events were dropped */
	TFREC_AMD_E_STATE = -5      /* call sequence error */
};

/* config.flags */
#define TFREC_AMD_F_ALL_FLUSHES 1u /* emit an event for EVERY decoder::flush call (parity/debug mode); default:
				      only flushes whose byte_cnt reaches the decoder's minimum length
				      (tfa1.cpp:49, tfa2.cpp:76/222, whb.cpp:484) */
#define TFREC_AMD_F_TIMING 2u      /* record HIP events around every kernel (tfrec_amd_get_timings) */
#define TFREC_AMD_F_INPUT_10X 8u   /* BASELINE config 5: the input is u8 IQ at 15.36 MS/s (TFREC_AMD_BLOCK_BYTES_10X bytes per
				      block and stream); a 10:1 decimating FIR in the reference's integer style (60 int16
				      taps, >>16 per tap, int16 store; defined in DESIGN.md, no reference counterpart)
				      produces the 1.536 MS/s int16 stream that enters downconvert::process_iq */
#define TFREC_AMD_F_BITS 16u        /* parity/debug: besides the flush events, report every bit the demodulators hand to
				      decoder::store_bit (decoder.h:39; tfa1.cpp:120, tfa2.cpp:281, whb.cpp:566) as BITS events: status
				      = TFREC_AMD_STATUS_BITS, byte_cnt = bits in this chunk (<= 512), rdata = the bits, LSB first,
				      seq = ordinal of the flush they precede, (end_sample, offset) = their order within that flush.
				      Window-parallel pipeline only. */
#define TFREC_AMD_F_SERIAL_CHAINS 4u /* run the demodulators as one serial lane per (stream, slot) -- the simple
				      GPU formulation kept as a cross-check of the window-parallel pipeline */

typedef struct {
	int32_t n_streams;   /* independent IQ streams in the batch (>=1) */
	int32_t types_mask;  /* bit n = sensor_e n as main.cpp -T: TFA_1 0x01, TFA_2 0x02, TFA_3 0x04, TX22 0x08, WHB 0x20 */
	int32_t thresh;      /* trigger threshold, main.cpp -t; 0 = the reference's auto mode (starts at 500, adapts by +-2
				every 4th block, fm_demod.cpp:23-27, 58-73) */
	int32_t filter_type; /* 0 = narrow, 1 = wide (-W), dsp_stuff.cpp:176-178 */
	int32_t device;      /* HIP device ordinal */
	int32_t max_blocks;  /* largest n_blocks a submit may carry (sizes the device buffers) */
	int32_t max_events;  /* device event buffer capacity per submit/drain cycle */
	uint32_t flags;      /* TFREC_AMD_F_* */
} tfrec_amd_config;

#define TFREC_AMD_STATUS_BITS 0x80 /* event.status of a BITS chunk (TFREC_AMD_F_BITS) */

/* One decoder::flush() call site (tfa1.cpp:180, tfa2.cpp:434, whb.cpp:696).  96 bytes. */
typedef struct {
	uint32_t stream;    /* stream index within the batch */
	uint8_t slot;       /* TFREC_AMD_SLOT_* */
	uint8_t status;     /* 1 = passes the decoder's CRC + sanity checks (a telegram), 2 = rejected, 0 = shorter than a telegram */
	uint16_t byte_cnt;  /* decoder byte_cnt at flush (saturated at 65535) */
	int32_t offset;     /* second flush() argument (tfa2.cpp:434; 0 for TFA_1 and WHB) */
	uint32_t seq;       /* ordinal of this flush within (stream, slot) since context creation */
	int64_t end_sample; /* decimated sample index (since stream start) at which flush fired */
	int64_t rssi_raw;   /* raw RSSI accumulator: tfa1.cpp:161, tfa2.cpp:373, whb.cpp:678 (an exact integer) */
	uint8_t rdata[64];  /* decoder rdata[0..64) at flush, before the decoder clears anything */
} tfrec_amd_event;

typedef struct tfrec_amd_ctx tfrec_amd_ctx;

/* kernel timings of the last submit (TFREC_AMD_F_TIMING), milliseconds */
typedef struct {
	float frontend_ms; /* (10:1 stage,) u8->s16 + 2-stage decimating FIR + trigger mask (+ auto-threshold pass) */
	float chains_ms;   /* all demodulator/decoder kernels together (end of front end -> end of last kernel) */
	float total_ms;    /* first kernel start to last kernel end */
	/* individual kernels of the window-parallel pipeline (0 when not run).  Window scan, then the TFA_2-family chain: */
	float windows_ms, spec_biquad_ms, repair_biquad_ms, fix_biquad_ms, slicer_ms, coop_slicer_ms, decode_ms, commit_ms;
	/* the WHB chain runs beside them on its own stream: its three biquad kernels together, then stage 2 */
	float whb_biquad_ms, whb_demod_ms, whb_decode_ms, whb_commit_ms; /* decode / commit: 0 (part of whb_demod) */
	/* ... and so does the TFA_1 chain (no biquad stage): short-window slicer, cooperative slicer, decode + commit */
	float tfa1_slicer_ms, tfa1_coop_slicer_ms, tfa1_decode_commit_ms;
	float fmdev_ms;    /* FM discriminator pass of the front end (tiles near trigger windows) */
	float whb_verify_ms; /* WHB stage 2 check: the exact decision-level recurrence, lane per stream (0: the exact stage 2 ran) */
} tfrec_amd_timings;

const char *tfrec_amd_version(void);
/* TFREC_AMD_FIFO_DEPTH of the loaded library (every submit in flight owns a full set of intermediate buffers: about
 * 7 GB per set and context at 1024 streams x 48 blocks -- several contexts on one device multiply that). */
int tfrec_amd_fifo_depth(void);
const char *tfrec_amd_strerror(int code);
/* text of the last HIP error seen by this thread ("" if none) */
const char *tfrec_amd_last_error(void);

/* Replaces, for cfg->n_streams receivers at once: `downconvert dc(2)` (engine.cpp:59), `fsk_demod fsk(&demods, thresh,
 * dbg)` (main.cpp:225) and the `new tfa1_demod / tfa2_demod / whb_demod` registrations of main.cpp:173-218
 * (types_mask = -T, thresh = -t, filter_type = -W). */
int tfrec_amd_create(const tfrec_amd_config *cfg, tfrec_amd_ctx **out);
int tfrec_amd_destroy(tfrec_amd_ctx *ctx);

/* Replaces n_blocks iterations of the block loop engine.cpp:67-86 for every stream: the u8 -> s16 conversion (:77-78),
 * downconvert::process_iq (:85, dsp_stuff.cpp:243-264) and fsk_demod::process (:86, fm_demod.cpp:34-74) with every
 * demodulator::start / ::demod and decoder::store_bit it drives (tfa1.cpp:120-190, tfa2.cpp:281-442, whb.cpp:566-707).
 * Process n_blocks 65536-byte blocks of every stream.  Stream s starts at d_iq + s*stream_stride_bytes
 * (device memory, u8 interleaved I,Q as the reference's -S dump files, sdr.cpp:233-234).  Asynchronous:
 * the work is ordered after what is already queued on hip_stream (a hipStream_t, NULL = default stream) -- the
 * producer of d_iq -- and runs on the context's own streams; d_iq must stay valid until the submit has been
 * drained (or tfrec_amd_sync returned).  All demodulator/decoder state carries over to the next submit exactly as
 * it carries from block to block in the reference.
 * A HIP failure in the middle of a submit (TFREC_AMD_E_HIP after work was enqueued) poisons the context: kernels may
 * already have advanced the carried state, so every later submit / drain returns TFREC_AMD_E_STATE until the context is
 * destroyed and recreated.  Argument errors (E_INVAL, a full FIFO) leave it untouched. */
int tfrec_amd_submit_device(tfrec_amd_ctx *ctx, const void *d_iq, size_t stream_stride_bytes, int n_blocks,
			    void *hip_stream);
/* Same with host memory: stages the batch through an internal device buffer (H2D copy included).  With pinned
 * memory (tfrec_amd_host_alloc) the copy is asynchronous and h_iq must stay untouched until the submit has been
 * drained; together with the submit/drain FIFO below this is the double-buffered feeder of SURVEY row f1: read batch
 * k+2 from disk while batch k+1 is copied/processed and batch k's events are dispatched. */
int tfrec_amd_submit_host(tfrec_amd_ctx *ctx, const uint8_t *h_iq, size_t stream_stride_bytes, int n_blocks);
/* Page-locked host memory for tfrec_amd_submit_host (NULL on failure). */
void *tfrec_amd_host_alloc(size_t bytes);
void tfrec_amd_host_free(void *p);

/* Wait for submitted work. */
int tfrec_amd_sync(tfrec_amd_ctx *ctx);

/* Replaces the calls `dec->flush(rssi, offset)` inside tfa1_demod::demod (tfa1.cpp:180), tfa2_demod::demod
 * (tfa2.cpp:434) and whb_demod::demod (whb.cpp:696): one tfrec_amd_event per call site and window.
 * Wait for the OLDEST submit that has not been drained yet, then copy its events to out[0..cap), ordered by
 * (stream, slot, seq).  *n_out = number written (0 if nothing was submitted).  Returns TFREC_AMD_E_OVERFLOW if the
 * device buffer or cap was too small (the events that fit are still returned; the others are lost -- the decoder state and
 * the flush ordinals `seq` move on, so a gap in seq shows where).
 * Submits and drains form a FIFO of depth TFREC_AMD_FIFO_DEPTH (4): a caller may queue submits k+1 .. k+3 before
 * draining submit k, so that the GPU works on them (front end of k+2, filter stage of k+1 and slicer/decoder stage of
 * k run beside each other, and the front end of k+3 is already queued when that of k+2 ends: with three, the front-end
 * stream idled from then until the host had drained k and submitted again) while the host copies and dispatches k's
 * events; one more undrained submit is refused with TFREC_AMD_E_STATE.  Every queued submit owns a full set of
 * intermediate buffers (~7 GB at 1024 streams x 48 blocks).  Alternating submit / drain behaves as one would expect. */
int tfrec_amd_drain_events(tfrec_amd_ctx *ctx, tfrec_amd_event *out, int cap, int *n_out);

/* Number of events of the oldest undrained submit (waits for it). */
int tfrec_amd_pending_events(tfrec_amd_ctx *ctx, int *n);

/* The dB value the reference demodulator passes to decoder::flush for this slot, computed with the
 * reference's host expressions (tfa1.cpp:180, tfa2.cpp:434, whb.cpp:696) including (int)(10*log10(0)). */
int tfrec_amd_rssi_db(int slot, int64_t rssi_raw);

/* Parity/debug (TFREC_AMD_F_INPUT_10X): the 1.536 MS/s int16 IQ the 10:1 stage produced for the last submit. */
int tfrec_amd_read_stage0(tfrec_amd_ctx *ctx, int stream, int16_t *out, size_t n_pairs);
/* Parity/debug: copy the decimated int16 IQ of the last submit for one stream (n_pairs*2 int16). */
int tfrec_amd_read_decimated(tfrec_amd_ctx *ctx, int stream, int16_t *out, size_t n_pairs);
/* fm_dev (dsp_stuff.cpp:284-292) is (int)(atan2(cj, cr) * 16384/pi) in double.  The device evaluates its own atan2
 * (4e-12 in the scaled angle); every sample whose scaled angle lies within 1e-9 of an integer is decided by an exact
 * slow path (double-double; the result the reference computes under a correctly rounded atan2), and the decisions are
 * logged (the first 62 per submit) and checked against THIS host's libm -- the arithmetic the reference binary uses
 * here -- when the submit is drained.  A run certifies itself: host_mismatch == 0 means every logged decision equals
 * the reference's; `undecidable` counts samples so close to a rounding midpoint of atan2 (< 0.06 ulp) that glibc's
 * documented error (0.55 ulp) lets the reference itself round either way (expected ~2e-13 per sample). */
typedef struct {
	uint64_t resolved;      /* samples decided by the slow path (of the submits drained so far) */
	uint64_t host_verified; /* ... of which were logged and compared with the host's libm */
	uint64_t host_mismatch; /* ... and differed from it */
	uint64_t undecidable;   /* slow-path samples within 0.06 ulp of an atan2 rounding midpoint */
	uint64_t reserved[4];
} tfrec_amd_fm_stats;
int tfrec_amd_get_fm_stats(tfrec_amd_ctx *ctx, tfrec_amd_fm_stats *out);
/* Samples decided by the slow path, including submits not drained yet (waits for them). */
int tfrec_amd_atan_uncertain(tfrec_amd_ctx *ctx, uint64_t *n);
/* Parity probe: the device's fm_dev on n 16-byte records -- kind 0: int32 quadruples (ar, aj, br, bj) = the arguments
 * of dsp_stuff.cpp:284; kind 1: int64 pairs (cr, cj) = its cross terms (:288-289), for directions no int16 quadruple
 * reaches; kind 2: the device's fm_dev_nrzs (dsp_stuff.cpp:269-279, with its +-1e9 clamp) on int32 quadruples.
 * out[n].  No context needed.  stats (may be NULL): as above, for this call. */
int tfrec_amd_fm_dev_probe(int device, int kind, const void *records, size_t n, int32_t *out, tfrec_amd_fm_stats *stats);
/* Parity probe: the device's iir2 (dsp_stuff.cpp:28-56: set(cutoff), then step() over in[0 .. n) from the zero state) ->
 * out[n], the outputs as doubles, bit for bit what the reference's normative build produces.  form 0:
 * the step as the reference associates it; form 1: the 3-multiply form the biquad passes and WHB stage 2 run (csrc/dsp_dev.h:
 * iir_step_t).  cutoff: iir2's argument, e.g. 0.5 / spb (tfa2.cpp:321), 2.0 / 64, 0.0025 / 64 (whb.cpp:610-611).  No context. */
int tfrec_amd_iir_probe(int device, double cutoff, int form, const double *in, size_t n, double *out);
int tfrec_amd_get_timings(tfrec_amd_ctx *ctx, tfrec_amd_timings *out);
/* Cumulative counters of the speculative stages (window-parallel pipeline only).  They only describe how the work
 * was done -- results do not depend on them. */
typedef struct {
	uint64_t biquad_segments;    /* biquad segments processed */
	uint64_t biquad_unconverged; /* parallel repair runs that reached the end of their segment without joining the
				        speculative trajectory (normal for a chain's short last segment) */
	uint64_t biquad_serial;      /* segments the chain walk had to repair serially */
	uint64_t tfa2_resliced;      /* tfa2 windows sliced again because the last_bit_idx assumption did not hold */
	uint64_t tfa1_recomputed;    /* 64-sample steps of long TFA_1 windows whose pre-computed peak detector piece did not
				        start from the true value and were recomputed */
	uint64_t biquad_repair_slots; /* 32-sample slots the first repair pass ran (a segment has up to 256: csrc/tfrec_dev.h kSegSlots) */
	uint64_t whb_respeculated;   /* (stream, submit) pairs whose lane-parallel WHB decision levels did not reproduce the exact
				        recurrence's decisions and were demodulated again by the exact kernel */
	uint64_t tfa1_scalar_groups; /* groups of 64 steps (4096 samples) of long TFA_1 windows that the lane-per-step cooperative slicer
				        left to its scalar walk (entered without a last_bit_idx and a candidate at a block's second
				        sample, a stale peak-detector piece, > 64 bits in a lane) */
	uint64_t tfa2_scalar_groups; /* ... of the TFA_2 family (more than 16 rounds of re-walking, entered with a block-relative 0) */
	uint64_t tfa1_vector_groups; /* groups the lane-per-step form did (counted by the experiments build of the library only, csrc/knobs.h;
				        0 otherwise: nearly every wave of the slicers would add to them): TFA_1, */
	uint64_t tfa2_vector_groups; /* TFA_2 family */
} tfrec_amd_stats;
int tfrec_amd_get_stats(tfrec_amd_ctx *ctx, tfrec_amd_stats *out);
/* Layout of the context's pipeline, named by its number of CHAIN streams: 6 = deep (default: the filter stage of submit
 * k+1 runs beside the slicer/decoder stage of submit k; plus the front-end stream and two low-priority streams for the
 * discriminator pass and the drain's copy), 4 = shallow (environment TFREC_AMD_DEEP=0 when the context is
 * created), 2 = the serial cross-check (TFREC_AMD_F_SERIAL_CHAINS).  Results do not depend on it.  No reference
 * counterpart. */
int tfrec_amd_get_layout(tfrec_amd_ctx *ctx, int *n_streams);
/* Device memory the context holds (front-end outputs, window tables, biquad outputs, state, event buffers: one set per
 * submit that may be in flight) and the page-locked host memory of its drain buffers, in bytes.  The caller's input
 * batches are not counted.  No reference counterpart. */
int tfrec_amd_get_memory(tfrec_amd_ctx *ctx, uint64_t *device_bytes, uint64_t *pinned_host_bytes);
/* Current trigger threshold of one stream (auto mode, fm_demod.cpp:58-73, moves it; fixed mode returns cfg.thresh). */
int tfrec_amd_read_thresh(tfrec_amd_ctx *ctx, int stream, int *thresh);

#ifdef __cplusplus
}
#endif
#endif
