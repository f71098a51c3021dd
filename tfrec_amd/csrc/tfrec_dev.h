// tfrec_amd/csrc/tfrec_dev.h -- device-side data layout shared by the kernels and the C-ABI glue.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/tfrec_amd.h"

// The kernels are written for ONE target: wave64 throughout, DPP row operations, v_pk_fma_f32 under a
// wave-wide rounding mode, v_dot2_i32_i16, v_permlane16/32_swap, hand-written GCN assembly.  Any other --offload-arch
// would fail in the assembler or, worse, compile to something else under wave32.
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__)
#error "tfrec_amd: the device code targets gfx950 (MI355X) only"
#endif
#if defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "tfrec_amd: the device code assumes 64-lane wavefronts"
#endif
#endif

namespace tfrec {

constexpr int kBlockDec = TFREC_AMD_BLOCK_DEC;  // 8192 decimated pairs per reference block
constexpr int kIndexSpan = 2 * kBlockDec;       // 'len' of fsk_demod::process / demodulator::start (fm_demod.cpp:38-40)
constexpr int kTailBytes = 112;                 // raw bytes of history carried between submits: 56 complex samples >= the
                                                // 46-sample halo of the 8+20 tap cascade (dsp_stuff.cpp:172-230) + 4 for
                                                // the previous decimated sample the discriminator needs, 16-byte multiple
constexpr int kNSlots = TFREC_AMD_NSLOTS;

// ---- front-end tile geometry: a 256-thread workgroup per tile, kFrontOut decimated outputs (and 2 * kFrontOut
// stage-1 outputs) per thread.  8 per thread: the per-wave scalar work (addresses, taps, edge tests) and the 6 raw
// samples a thread's stage-1 group shares with its neighbour are paid per 8 outputs instead of per 4.
#ifndef TFREC_AMD_FRONT_OUT
#define TFREC_AMD_FRONT_OUT 8
#endif
constexpr int kFrontOut = TFREC_AMD_FRONT_OUT;      // stage-2 outputs per thread: 4 or 8
static_assert(kFrontOut == 4 || kFrontOut == 8, "the trigger-mask packing knows nibbles and bytes");
constexpr int kFrontThreads = 256;
constexpr int kTileDec = kFrontThreads * kFrontOut;  // decimated outputs per workgroup tile
// LDS image of the stage-1 outputs ((I, Q) float pairs, 8 bytes a slot): a thread's group of 2 * kFrontOut slots is
// followed by kY1Pad pad slots, so that the 16-byte reads and writes of neighbouring lanes (a lane stride of 64 or 128
// bytes would put every 2nd / 4th lane on the same banks) fall on different banks.
constexpr int kY1Group = 2 * kFrontOut;
#ifndef TFREC_AMD_Y1_PAD
#define TFREC_AMD_Y1_PAD 2
#endif
constexpr int kY1Pad = TFREC_AMD_Y1_PAD;
constexpr int kY1Stride = kY1Group + kY1Pad;
__host__ __device__ constexpr int y1_phys(int slot) { return slot + kY1Pad * (slot / kY1Group); }
constexpr int kY1Count = y1_phys(2 * kTileDec + 24) + 8;  // stage-1 outputs held per tile (need 2*T+24)
// the discriminator pass keeps the 256 x 4 geometry
constexpr int kFmThreads = 256, kFmTile = 1024;

// Second-stage taps as h / 65536 (exact in fp32): the multiplier of the FMA form of the FIR stages (frontend.hip).
struct FrontTaps {
	float f2[20][2];  // stage-2 taps / 65536, each twice: the (I, Q) operand of a packed FMA as it sits in a scalar register pair
};

// ---- biquad (dsp_stuff.cpp:28-56); state as the reference's members, coefficients in FrontParams
struct Biquad {
	double dn1, dn2, yn, yn1;
};

struct BiquadCoef {
	double b0, b1, b2, a1, a2;
};

// ---- persistent per-(stream, slot) demodulator + decoder state.  Field names follow the reference members
// (tfa1.h:25-33, tfa2.h:30-43, whb.h:44-60, decoder.h:49-57).
struct ChainState {
	// demodulator
	int32_t last_bit_idx;  // decoder.h:72 (block-relative int16 index units, rebased by demodulator::start)
	int32_t timeout_cnt;
	int32_t mark_lvl;      // tfa1
	int32_t rssi_i;        // tfa1 / tfa2 integer rssi
	int32_t bitcnt, dmin, dmax, offset, last_bit;  // tfa2
	int32_t last_dev, avg_of;                      // whb
	int32_t prev_i, prev_q;                        // last_i/last_q: previous decimated sample
	uint64_t step, last_peak;                      // whb
	double rssi_d;                                 // whb
	Biquad iir, iir_avg;
	// decoder
	uint32_t sr;
	int32_t sr_cnt, byte_cnt, invert, synced;
	int32_t w_last_bit, psk, last_psk, nrzs;  // whb descrambler chain
	uint32_t lfsr;
	uint32_t seq;  // flush ordinal
	uint32_t pad_;
	uint8_t rdata[256];
};

struct ChainParams {
	int32_t kind;        // 0 tfa1, 1 tfa2 family, 2 whb
	int32_t sensor_type; // sensor_e value (decoder.h:11-19)
	int32_t window;      // timeout reload: 400 / (int)(16*spb) / (int)(8*spb)
	int32_t min_bytes;   // smallest byte_cnt a flush can turn into a telegram
	double spb;
	BiquadCoef iir, iir_avg;
	// tfa2.cpp:397 numbits = (int)(((tdiff / 2) + spb / 2) / spb) without the fp64 division: for every h = tdiff / 2 the
	// slicer can pass (tdiff < 32 * spb) it equals (h * nb_mul + 2^39) >> 40 -- checked exhaustively against the fp64
	// expression when the context is created; 0 = no multiplier passed the check, the kernels divide.
	uint64_t nb_mul;
	// tfa2.cpp:393 "tdiff > spb / 4 && tdiff < 32 * spb" for the integer tdiff: td_lo <= tdiff <= td_hi
	int32_t td_lo, td_hi;
};

__host__ __device__ inline int tfa2_numbits_mul(int tdiff, uint64_t nb_mul)
{
	return (int)(((uint64_t)(uint32_t)(tdiff >> 1) * nb_mul + (1ull << 39)) >> 40);
}

// everything one chains launch needs, passed by value as kernel argument
struct ChainLaunch {
	int32_t n_active;
	int32_t slot[kNSlots];
	ChainState *states[kNSlots];
	ChainParams params[kNSlots];
};

// fsk_demod state of one stream in auto-threshold mode (fm_demod.h:23-30, fm_demod.cpp:18-32)
struct FskState {
	int32_t thresh;         // current trigger threshold
	int32_t triggered_avg;  // fm_demod.cpp:58
	int32_t runs;           // blocks processed so far (fm_demod.cpp:37)
	int32_t last_trig;      // last sample with pwr > thresh, relative to the start of the current submit
};

// fm_dev samples decided by the exact slow path (fm_resolve.h): logged so that the host can check them against its
// own libm when the batch is drained (capi.hip)
struct FmLogEntry {
	double cr, cj;   // the discriminator's cross terms (exact integers)
	int32_t result;  // what the device returned
	float margin;    // |theta - rounding midpoint| in ulps of the angle (fm_dev_resolve)
};
constexpr int kFmLogCap = 62;
constexpr double kFmUndecidableUlps = 0.06;  // glibc's atan2 is within 0.55 ulp: closer to a midpoint it may round either way

struct EventBuf {
	uint32_t count;     // events appended (may exceed capacity -> overflow)
	uint32_t capacity;
	unsigned long long uncertain;  // fm_dev results within 1e-9 of a truncation boundary: decided by the exact slow path
	uint32_t fm_logged;            // ... of which the first kFmLogCap are in fm_log
	uint32_t fm_undecidable;       // ... whose margin is below kFmUndecidableUlps
	FmLogEntry fm_log[kFmLogCap];
	// fmdev_kernel defers its flagged samples to fm_resolve_kernel (the slow path as a call inside the hot kernel
	// doubled its registers: 0.8 ms per batch): (stream << 32 | sample) of every flagged sample; more than
	// kFmListCap pending = the resolve kernel rescans the whole submit instead
	uint32_t fm_pending;
	uint32_t dead;  // events marked TFREC_AMD_STATUS_DEAD (a WHB stream's speculative events, replaced by the exact kernel's)
	unsigned long long fm_list[256];
};
constexpr int kFmListCap = 256;

// ---- window-parallel pipeline (chains2.hip)

// Result of running one trigger window of a TFA_1 / TFA_2-family demodulator (one lane per window).
struct WinResult {
	int32_t nbits;         // bits written to the window's bit region
	int32_t closed;        // the window's flush fired inside this submit
	int32_t rssi_i;        // raw rssi at flush (or so far, if the window is still open)
	int32_t offset;        // tfa2 offset at flush (or so far)
	int32_t lbi_out;       // last_bit_idx after the window, relative to the block of its last sample
	int32_t first_cand_g;  // tfa2: first sample whose slicer saw a candidate edge (-1: none) -- speculation check
	int32_t bitcnt, dmin, dmax, last_bit, mark_lvl;  // slicer state (needed when the window stays open)
	int32_t resume;        // long windows: first 32-sample slot the cooperative slicer still has to do (-1: none)
};

// decoder::store_bit over ONE window's bits (TFA_1 / TFA_2 family), done window-parallel by decode_kernel
struct WinDecode {
	uint32_t sr;       // decoder registers after the window's last bit (before the flush)
	int32_t sr_cnt, byte_cnt, invert;
	int32_t wlen;      // TFA: rdata[0 .. wlen) were (re)written by this window (window 0 of a chain: all 64)
	uint32_t lfsr;     // WHB: descrambler history after the window; `invert` then holds kWhbF* flags
	unsigned long long wmask;  // WHB: bit b = rdata[b] was written by this window (window 0: all ones)
	uint8_t vals[64];  // rdata[0 .. 64) as the window leaves them (valid where written)
};
constexpr int kWhbFPsk = 1, kWhbFSynced = 2, kWhbFLastBit = 4, kWhbFNrzs = 8;
static_assert(sizeof(WinDecode) == 96, "WinDecode layout");

// whb_decoder registers at the first bit of a window, as tracked by whb_demod_kernel (window-parallel replay)
struct WhbStart {
	uint32_t sr, lfsr;
	int32_t sr_cnt, byte_cnt, synced;
	int32_t pad_[3];
};

// One record per 64-sample step in which whb_demod_kernel<false> ran the decision-level average (and one per window that
// began with the decoder locked, and an end mark): everything whb_verify_kernel needs to walk a stream's submit as ONE
// flat sequence with its loads queued several steps ahead -- where the step's stage-1 outputs are, on how many of them the
// filter ran, the speculated decisions, and what the window froze if the decoder locked in this step.
struct WhbStepRec {
	unsigned long long below;  // bit k: the demodulator took "dev(k) < (int)avg(k)" (whb.cpp:662) as true
	uint32_t meta;             // kWhbRec*: 32-sample slot of the step's first stage-1 output in the stream's dev32 row | (samples - 1) << 22 | flags
	int32_t avgf;              // kWhbRecLock: the integer the demodulator froze (whb.cpp:653-654)
};
constexpr uint32_t kWhbRecOffMask = 0x3fffffu;  // offset in 32-sample slots (steps start on slot boundaries): 2^22 slots = 4096 blocks
constexpr int kWhbRecNvShift = 22;              // 6 bits: samples of the step on which the filter ran, minus one
constexpr uint32_t kWhbRecLock = 1u << 28;      // the decoder locked on the step's last filtered sample
constexpr uint32_t kWhbRecAmb = 1u << 29;       // a candidate test of the window against the frozen average is ambiguous
constexpr uint32_t kWhbRecClosed = 1u << 30;    // the window's flush fired in this submit
constexpr uint32_t kWhbRecPseudo = 1u << 31;    // no filter step: a window that began (and stayed) locked
constexpr uint32_t kWhbRecEnd = 0xffffffffu;    // no more records
constexpr int kWhbRecSlack = 16;               // records whb_verify_kernel may read past the end mark (never interpreted)
static_assert(sizeof(WhbStepRec) == 16, "one 16-byte load per record");

// iir_avg of whb_demod (whb.cpp:611, 654) as whb_verify_kernel carries it: the last two outputs, bit for bit, and the
// last two inputs (0.5 * a stage-1 output each) as the integers
struct WhbExact {
	double y1, y2;
	int32_t fd1, fd2;
	int32_t carry;   // whbx0 only: exact minus speculated frozen average of a locked window open at the submit's start
	int32_t pad_;
};

// TFA_1 peak detector (tfa1.cpp:157-160) over one piece of kMarkSlots slots of a long window, run by mark_kernel
// from a warm-up (the detector forgets its state at every new peak): valid iff start == the true value
struct MarkPiece {
	int32_t start;  // mark_lvl before the piece's first sample, as the warm-up produced it
	int32_t end;    // mark_lvl after the piece's last sample
	int32_t max;    // max of mark_lvl over the piece (rssi, tfa1.cpp:161-162)
	int32_t pad_;
};
constexpr int kMarkSlots = 32;   // 1024 samples per piece
constexpr int kMarkWarmSlots = 8;  // 256 samples of warm-up

// full biquad state at the end of a window (speculative or repaired run)
struct BiquadEnd {
	double dn1, dn2, yn, yn1;
};

struct WorkQueue {
	uint32_t count;  // items pushed
	uint32_t head;   // items taken
	uint32_t head2;  // items taken by a second pass over the same queue
	uint32_t head3;  // ... and by a third
};

struct WinTables {
	int32_t cap;            // windows per chain the tables can hold
	int32_t bit_words;      // 32-bit words of bit storage per chain
	int32_t *count;         // [chains] windows found in this submit
	int32_t *cont;          // [chains] window 0 continues a window left open by the previous submit
	int32_t *timeout_next;  // [chains] timeout_cnt after the last sample of this submit
	int32_t *open;          // [chains*cap] first sample of the window
	int32_t *close;         // [chains*cap] sample at which the window's flush fires (>= M: after this submit)
	WinResult *result;      // [chains*cap]
	WinDecode *decode;      // [chains*cap]
	WhbStart *whbstart;     // [n_streams*cap]
	uint32_t *cand;         // [n_streams*slots] TFA_1 long windows: "dev < mark_lvl/2" bits of the slot's samples
	MarkPiece *mark;        // [n_streams*slots] indexed by the piece's first slot
	uint32_t *bits;         // [chains*bit_words] emitted bits, LSB first; window j of a chain starts at word (open>>6)+3*j
	uint2 *items;           // [8][chains*cap] work items; slicer queues 2*kind + {0: long, 1: short windows}: (chain, j);
	                        // queues 4 (TFA_2 family) and 6 (WHB): biquad segments (chain, segment); queue 5: unused;
	                        // queue 7: peak-detector pieces of long TFA_1 windows (chain, j | p << 17)
	WorkQueue *queue;       // [8]
	int32_t slots;          // 32-sample slots per chain row of the window-relative arrays below
	double2 *ckpt;          // [(chains - ck_c0)*slots] (yn, yn1) after the last sample of each slot, speculative biquad run
	int32_t ck_c0;          // first chain with a biquad stage (TFA_1, registered first, has none): rows of ckpt start there
	int32_t ld_c0;          // first TFA_2-family chain: rows of ld16 start there (the family's slots are adjacent)
	// biquad segments: the in-window slots of a chain, numbered consecutively across windows ("virtual slots"),
	// are cut into segments of kSegSlots slots
	int32_t segcap;         // segments per chain the tables can hold
	uint2 *segstart;        // [chains*segcap] (window j, window-relative slot) where the segment starts
	int32_t *vtotal;        // [chains] virtual slots of the chain in this submit
	BiquadEnd *segend1;     // [chains*segcap] state after the segment, speculative run
	BiquadEnd *segend2;     // [chains*segcap] state after the segment, repair run that did not converge
	int32_t *segfix;        // [chains*segcap] repair run: slots rewritten | kSegConverged
	BiquadEnd *segend3;     // [chains*segcap] ... second repair run (started from segend2 of the segment before)
	int32_t *segfix2;       // [chains*segcap] second repair run: slots rewritten | kSegConverged | kSegRan (0: not run)
	int32_t *overflow;      // set when a chain found more than cap windows
	unsigned long long *stats;  // [8] tfrec_amd_stats
	const uint32_t *prevdec;  // [n_streams] the decimated sample before this submit's first one (front end)
	int32_t *timeout_carry; // [chains] timeout_cnt the window scan carries from submit to submit (ONE array per context,
	                        // shared by the two table sets: the scan of submit k+1 must not wait for the chains of k)
	// WHB stage 2, speculate + verify (chains2.hip K4'): per 64-sample step in which the decision-level average ran,
	// the decisions "dev < (int)avg" the demodulator kernel took from its lane-parallel evaluation of the filter
	WhbStepRec *whbrec;          // [n_streams * whbrec_stride], in the order whb_verify_kernel walks them
	int32_t whbrec_stride;       // records a stream can have in one submit (filter steps M / 64 + cap, one per window that
	                             // begins locked, the end mark) + the slack whb_verify_kernel's prefetch reads ahead
	WhbExact *whbx;              // [n_streams] the filter's exact state, carried by whb_verify_kernel (ONE array per context)
	int32_t *whbfail;            // [n_streams] set by whb_verify_kernel: the stream's speculation failed in this submit
	// ... and what the exact kernel needs to do such a stream's submit again:
	ChainState *whbsnap;         // [n_streams] the WHB chain state whb_demod_kernel<false> started this submit from
	WhbExact *whbx0;             // [n_streams] the exact filter state whb_verify_kernel started this submit from
	uint32_t *whbseen;           // [n_streams] whbgen[s] as whb_demod_kernel<false> saw it before it read the state
	uint32_t *whbgen;            // [n_streams] redone submits of the stream so far (ONE array per context)
	ChainState *whbX;            // [n_streams] the chain state after the stream's last redone submit (ONE array per context)
	// ... round 6 (whb_check.h): the check as an exact chain PER LANE.  whb_demod_kernel<false> also writes the filter's INPUT SEQUENCE
	// of the stream -- the stage-1 output of every sample the decision-level average ran on, in order, whatever window and step
	// it came from (the filter does not know about either: it pauses while the decoder is locked and goes on where it stopped) --
	// and whb_chain_kernel walks it 64 inputs a round, a stream per lane:
	int32_t *whbdense;             // [n_streams * whbdense_stride] the sequence
	int32_t whbdense_stride;       // M + 64
	int32_t *whbdense_n;           // [n_streams] its length in this submit
	unsigned long long *whbxbits;  // [n_streams * whbx_stride] round r: bit k = the exact "dev < (int)avg" (whb.cpp:654, 662) of input 64 r + k
	double2 *whbxsnap;             // [n_streams * whbx_stride] round r: (y, y one input earlier) after input 64 r + 63
	int32_t whbx_stride;           // rounds a stream can have in one submit + 2
	int32_t whb_force_fail;      // tests (TFREC_AMD_WHB_FORCE_FAIL=N): declare every N-th (stream + submit) failed
	int32_t whb_submit_seq;
	// The redo works on a PRIVATE copy of the stream's chain state (whbscr, ONE array per context: redos run one after the
	// other on vx) -- the speculative kernels of the submits behind it work in place on L.states meanwhile -- and publishes
	// the result to whbX and to the live state (whbpub = L.states[whb slot], set for the redo launch only) at its end.
	ChainState *whbscr;
	ChainState *whbpub;
	// tests (TFREC_AMD_WHB_TEST_PERTURB=D): the speculative kernel freezes (int)avg + D when the decoder locks, and "off by
	// one" becomes "off by at most |D|" in the ambiguity rule and in the check: drives the carry / amb / redo paths of a
	// window that spans submits with ordinary input.  0 in production (tolerance 1).
	int32_t whb_test_perturb;
	int32_t tfa2_vec;            // ... and so does the TFA_2 family's, once a window's thresholds are frozen (TFREC_AMD_TFA2_VEC=0: the scalar walk)
	int32_t tfa1_vec;            // the cooperative TFA_1 slicer walks 64 steps at a time, a step per lane (TFREC_AMD_TFA1_VEC=0: the scalar walk)
};
constexpr int kStatusDead = 0xff;  // tfrec_amd_event::status of a retracted event: never reported

// Streams and events of one submit of the window-parallel pipeline.  Every (protocol family, stage) pair owns a
// stream, so that stage A (biquads) of submit k+1 runs beside stage B (slicers, decoders) of submit k; the two
// submits use different table / buffer sets.  k2 == cs and kw == aux is allowed (shallow layout: with the HIP
// default of 4 hardware queues more streams would share queues and serialise).
struct PipeCtl {
	hipStream_t fs;            // front-end stream
	hipStream_t ws;            // stream of the window scan: fs, or (deep layout) kw, after ev_front
	hipEvent_t ev_front;       // front end done (fs)
	hipStream_t k2, kw;        // stage A: biquads of the TFA_2 family / of WHB
	hipStream_t cs, aux, t1;   // stage B: TFA_2 family, WHB, TFA_1 (no stage A)
	hipStream_t vx;            // stage C of WHB: whb_verify_kernel (== aux in the shallow layout)
	hipEvent_t ev_aux;         // whb_demod_kernel done (aux)
	int *whb_carry;            // [n_streams] whb_verify_kernel: exact minus speculated frozen average of a window still open
	hipEvent_t ev_win;         // window scan done (fs)
	hipEvent_t ev_fork;        // first TFA_2 biquad pass done (k2): TFA_1 starts
	hipEvent_t ev_k2, ev_kw;   // stage A done
	hipEvent_t ev_fm;          // the discriminator pass done, when it runs at the head of kw instead of k2
	// TFA_2 family, stage B split: once the long windows' heads are sliced (cs), the cooperative slicers of their tails
	// run on cz beside the short windows' slicers on cs (nullptr: one after the other on cs)
	hipStream_t cz;
	hipStream_t fq;            // the discriminator pass's own stream (TFREC_AMD_FMDEV_OWN), or nullptr: at the head of k2
	hipStream_t ks;            // the speculative pass of the TFA_2 family's biquads (TFREC_AMD_SPEC_OWN), or nullptr: at the head of k2
	hipEvent_t ev_spec;        // ... done (ks): the repair passes on k2 start
	hipEvent_t ev_heads, ev_coop;
	hipEvent_t done[3];        // end of the submit on cs / aux / t1
	hipEvent_t *tev;           // optional timing marks (kTimingMarks)
	// the FM discriminator pass, when it runs at the head of stage A of the TFA_2 family (k2) instead of behind the
	// front end (its only consumer is that stage): wmax > 0
	int fmdev_wmax;
	double fm_flag_eps;  // distance of the scaled angle to an integer below which a sample goes to the exact slow path
	int16_t *fmdev_out;
	const uint32_t *prevdec;
};
constexpr int kTimingMarks = 30;

constexpr int kNQueues = 8;
// one more counter after the work queues, with a (stream, slot) list behind the queues' items: the TFA_2-family
// chains whose commit found a window sliced under a wrong last_bit_idx assumption (commit_wave_kernel takes them)
constexpr int kDeferQueue = kNQueues;
#ifndef TFREC_AMD_SEG_SLOTS
#define TFREC_AMD_SEG_SLOTS 256
#endif
constexpr int kSegSlots = TFREC_AMD_SEG_SLOTS;  // biquad segments: 256 in-window slots (>= 7400 samples: windows are >= 11 slots long).  Round 5
                                                // (profiles/r05_ab_segments.txt): 128 -> 256 with the same number of waves = a quarter fewer repair
                                                // slots, the batch 2.5 % shorter; 512: no better (the passes stretch), 1024: 17 % worse
constexpr int kSegConverged = 0x40000000, kSegRan = 0x20000000;
constexpr int kLongWindow = 1024;  // samples; longer windows go to the wave-cooperative slicers (default; TFREC_AMD_COOP_MIN).
                                   // Until round 5 the cooperative slicers were scalar walks (51 scalar instructions per
                                   // accepted edge) and 4096 measured 4-8 % better than 2048; with a step per lane they cost
                                   // less per window than a lane of the lane-per-window kernels, whose longest window sets
                                   // their duration: 1024-2048 measure 1.5 % better than 4096 (profiles/r05_ab_coop_min.txt)

static_assert(sizeof(tfrec_amd_event) == 96, "event ABI is 96 bytes");
static_assert(offsetof(tfrec_amd_event, rdata) == 32, "event rdata offset");
static_assert(offsetof(ChainState, rdata) % 16 == 0 && sizeof(ChainState) % 16 == 0, "ChainState rdata must be 16-byte aligned");

}  // namespace tfrec
