// tfrec_amd/csrc/capi.hip -- C ABI (include/tfrec_amd.h): context, submit, drain.  No torch types.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <vector>

#include "knobs.h"
#include "tfrec_dev.h"

namespace tfrec {
hipError_t launch_decim10(hipStream_t st, const uint8_t *iq, size_t stride, int n_streams, int n_blocks,
			  const uint8_t *tail_in, uint8_t *tail_out, uint32_t *out, size_t out_stride);
hipError_t launch_frontend(hipStream_t st, const uint8_t *iq, size_t stride, int n_streams, int n_blocks,
			   const uint8_t *tail_in, uint8_t *tail_out, uint32_t *dec, size_t dec_stride,
			   unsigned long long *mask, size_t mask_stride, uint32_t *prevdec, int thresh, const FrontTaps &taps,
			   bool in16);
hipError_t launch_fmdev(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			size_t mask_stride, const uint32_t *prevdec, int16_t *fmdev, size_t fmdev_stride, EventBuf *eb,
			int n_streams, int n_blocks, int wmax, double flag_eps);
hipError_t launch_pipeline(const PipeCtl &P, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			   size_t mask_stride, const int16_t *fmdev, size_t fmdev_stride, int n_streams, int n_blocks,
			   long long sample_base, const ChainLaunch &L, const WinTables &T, int16_t *ld16, int32_t *dev32,
			   tfrec_amd_event *events, EventBuf *eb, uint32_t flags);
hipError_t launch_fm_probe(hipStream_t st, const int32_t *quads, size_t n, int32_t *out, EventBuf *eb, int kind);
hipError_t launch_iir_probe(hipStream_t st, const double *in, size_t n, const BiquadCoef &c, double *out, int form);
hipError_t launch_threshold(hipStream_t st, const uint32_t *dec, size_t dec_stride, unsigned long long *mask,
			    size_t mask_stride, int n_streams, int n_blocks, FskState *fsk, int wmax);
hipError_t launch_chains(hipStream_t st, const uint32_t *dec, size_t dec_stride, const unsigned long long *mask,
			 size_t mask_stride, int n_streams, int n_blocks, long long sample_base, const ChainLaunch &L,
			 tfrec_amd_event *events, EventBuf *eb, uint32_t flags);
}  // namespace tfrec

using namespace tfrec;

// Buffer / table sets = submits that may be in flight (the FIFO depth): front end of submit k+2, biquad stage of
// k+1 and slicer stage of k run beside each other in the deep layout
constexpr int kSets = TFREC_AMD_FIFO_DEPTH;
// header of a set's event block: EventBuf + 16 bytes (the window tables' overflow flag, at kEvOverflowOff on the device and
// in the host copy alike), padded
constexpr size_t kEvOverflowOff = sizeof(EventBuf);
constexpr size_t kEvFreshBytes = kEvOverflowOff + 16;  // what a submit resets from d_eb_fresh: EventBuf + the flag
constexpr size_t kEvHeader = (kEvFreshBytes + 255) & ~(size_t)255;

static thread_local char g_err[256] = "";

static int hip_fail(hipError_t e, const char *what)
{
	snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
	return TFREC_AMD_E_HIP;
}
#define HIPCHK(call)                                   \
	do {                                           \
		hipError_t e_ = (call);                \
		if (e_ != hipSuccess)                  \
			return hip_fail(e_, #call);    \
	} while (0)

struct FmTotals {
	unsigned long long resolved = 0, verified = 0, mismatch = 0, undecidable = 0;
};

struct tfrec_amd_ctx {
	tfrec_amd_config cfg;
	ChainLaunch launch;
	FrontTaps taps;
	// front-end outputs, one set per submit in flight like the event buffers: the front end of submit k+2 (its own stream)
	// runs beside the demodulator chains of submit k
	uint32_t *d_dec[kSets] = {};
	size_t dec_stride = 0;  // uint32 units
	unsigned long long *d_mask[kSets] = {};
	size_t mask_stride = 0;
	int16_t *d_fmdev[kSets] = {};  // [n_streams][m_max] fm_dev of the decimated samples (computed near windows)
	uint32_t *d_prevdec[kSets] = {};  // [n_streams] the decimated sample before the submit's first one
	bool need_fmdev = false;                      // a TFA_2-family demodulator is registered
	// HIP multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams
	// that share a queue serialise (profiles/ubench/queues.hip).  The deep layout (stage A of submit k+1 beside stage B
	// of submit k) has six streams in two priority classes (see cp below); the shallow one has k2 = cs and kw = aux.
	hipStream_t fs = nullptr;                     // front-end stream + window scan (+ the drain's device-to-host copies)
	hipStream_t cs = nullptr;                     // TFA_2-family slicers and decoders (the caller's stream only orders the input)
	hipStream_t k2 = nullptr, kw = nullptr;       // biquad stages (deep layout only: else aliases of cs / aux)
	// The drain's device-to-host copies must not queue behind the work of younger submits, so they need a hardware
	// queue of their own.  The runtime keeps one pool of hardware queues PER PRIORITY: fs / cs / aux / t1 are the
	// high-priority streams, and cp (with k2 / kw in the deep layout) has normal priority -- no pool holds more than
	// four streams, whatever GPU_MAX_HW_QUEUES is.
	hipStream_t cp = nullptr;
	bool deep = false;
	bool scan_on_kw = false;                      // deep layout: the window scan runs at the head of kw, not on fs
	bool fmdev_k2 = false;                        // the discriminator pass runs at the head of k2, not behind the front end
	hipEvent_t ev_in[kSets] = {}, ev_front[kSets] = {};
	hipEvent_t ev_pipe[kSets][8] = {};
	hipStream_t fq = nullptr;                    // PipeCtl::fq (TFREC_AMD_FMDEV_OWN)
	hipStream_t ks = nullptr;                    // PipeCtl::ks (TFREC_AMD_SPEC_OWN)
	hipStream_t cq = nullptr;                    // the drain's copies, when not on cp (TFREC_AMD_COPY_OWN)
	hipStream_t cz = nullptr;                    // PipeCtl::cz (TFREC_AMD_COOP_STREAM)                // per set: window scan done, TFA_1 fork, TFA_2 / WHB biquads done
	int last_set = 0;
	// one set per submit in flight, like the front-end outputs: the window scan and the biquads of submit k+1 fill
	// theirs while the slicers of submit k still read the other
	int16_t *d_ld16[kSets] = {};   // [chains][m_max] tfa2-family biquad outputs
	int32_t *d_dev32[kSets] = {};  // [n_streams][m_max] WHB stage-1 outputs
	WinTables win[kSets] = {};
	void *win_block[kSets] = {};
	int32_t *d_tcarry = nullptr;                 // WinTables::timeout_carry
	WhbExact *d_whbx = nullptr;                  // WinTables::whbx
	int *d_whbcarry = nullptr;                   // PipeCtl::whb_carry
	uint32_t *d_whbgen = nullptr;                // WinTables::whbgen
	ChainState *d_whbX = nullptr;                // WinTables::whbX
	ChainState *d_whbscr = nullptr;              // WinTables::whbscr
	int whb_test_perturb = 0;                    // TFREC_AMD_WHB_TEST_PERTURB (tests)
	int whb_force_fail = 0;                      // TFREC_AMD_WHB_FORCE_FAIL (tests)
	int submit_seq = 0;
	hipStream_t vx = nullptr;                    // whb_verify_kernel: an alias of cp (deep layout) or of aux
	hipEvent_t ev_aux[kSets] = {};               // whb_demod_kernel of the set's submit done
	FskState *d_fsk = nullptr;  // auto-threshold mode only
	int wmax = 0;
	uint8_t *d_tail[kSets] = {};
	int tail_sel = 0;
	// TFREC_AMD_F_INPUT_10X: output of the 10:1 stage (1.536 MS/s int16 pairs, one buffer per set) and its raw history
	uint32_t *d_in16[kSets] = {};
	size_t in16_stride = 0;  // uint32 units
	uint8_t *d_tail10[kSets] = {};
	bool in10x = false;
	// One event buffer set per submit in flight (FIFO of depth TFREC_AMD_FIFO_DEPTH): submits may be queued while the
	// host still drains an older one
	tfrec_amd_event *d_events[kSets] = {};
	EventBuf *d_eb[kSets] = {};
	uint8_t *d_evblock[kSets] = {}, *h_evblock[kSets] = {};  // what d_eb / d_events and h_eb / h_events point into
	uint8_t *h_evblock_dev[kSets] = {};                      // the page-locked blocks as the device addresses them (drain_copy_kernel)
	EventBuf *d_eb_fresh = nullptr;       // { 0, max_events, 0 }: copied over a set's EventBuf when a submit starts
	// Pinned staging for the drain, one per set: the device-to-host copies of a submit's event buffer are queued on cp
	// when the submit is made (behind its three end-of-chain events), so they are done when the host comes to drain it.
	// The number of events is not known then: `copy_guess` of them are copied ahead (twice the last submit's count), the
	// drain fetches the rest if there are more.
	tfrec_amd_event *h_events[kSets] = {};
	EventBuf *h_eb[kSets] = {};
	hipEvent_t copied[kSets] = {};
	uint32_t copied_n[kSets] = {};
	uint32_t copy_guess = 4096, copy_guess_min = 4096;  // TFREC_AMD_COPY_GUESS_MIN (tests: exercise the fetch-the-rest path)
	std::vector<uint32_t> sort_idx, sort_start;
	hipEvent_t done[kSets][3] = {};  // end of the submit that owns the set, on the cs / aux / t1 stream
	int head = 0, inflight = 0;           // oldest undrained set, submits not yet drained (0..TFREC_AMD_FIFO_DEPTH)
	int last_drained = -1;
	uint8_t *d_stage[kSets] = {};  // tfrec_amd_submit_host: device staging, one per buffer set
	size_t stage_bytes[kSets] = {};
	long long sample_base = 0;
	int last_blocks = 0;
	hipEvent_t ev[kSets][4] = {};  // start, after the front end, end of the submit, after the discriminator pass
	hipStream_t aux = nullptr;  // second stream: WHB stage 2 runs beside the TFA slicers
	hipEvent_t tev[kSets][kTimingMarks] = {};
	hipStream_t t1 = nullptr;  // TFA_1 slicer chain (needs no biquad stage: runs beside the TFA_2-family biquads)
	bool whb_active = false;
	bool timed = false;
	// fm_dev samples decided by the exact slow path: all / checked against this host's libm at drain / differing from
	// it / closer to a rounding midpoint than glibc's error bound
	FmTotals fm;
	// fm_dev samples closer than this to a truncation boundary take the exact slow path.  1e-9 = 250x the fast path's
	// error bound; TFREC_AMD_FM_FLAG_EPS (tests) widens it to drive the slow path -- exact for any value -- through the
	// pipeline with ordinary input: 1e-3 fills the deferred list, 0.6 overflows it (every sample: the rescan path)
	double fm_flag_eps = 1e-9;
	// A HIP call failed in the middle of a submit: kernels of it may already have run on carried state (FIR history, chain
	// state, the FIFO's bookkeeping), so the context cannot continue exactly.  Every later submit / drain returns
	// TFREC_AMD_E_STATE; destroy and recreate.
	bool poisoned = false;
	size_t dev_bytes = 0, pinned_bytes = 0;  // tfrec_amd_get_memory
	// TFREC_AMD_HOST_PROF=1: host-side time of the submit / drain calls, printed when the context is destroyed
	double hp_submit = 0, hp_wait = 0, hp_copy = 0, hp_sort = 0, hp_gap = 0, hp_lat = 0, hp_s2s = 0;
	long hp_n = 0, hp_gap_n = 0;
};

namespace tfrec {
// The drain's device-to-host copy as a kernel of our own (16 bytes per lane into the page-locked block, which the device addresses
// directly).  hipMemcpyAsync did the same with the runtime's copy kernel -- but two or three times after every synchronize (the 6th and
// 7th submit of the driver's 20-step line) the CALL blocked the host for a whole batch period, now and then for two (13 ms: the pipeline
// ran dry, 6.2 instead of 5.75 ms per step): profiles/r06_host_stalls.txt.
__global__ __launch_bounds__(256) void drain_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
		dst[i] = src[i];
}
}  // namespace tfrec

namespace {
struct PoisonGuard {
	tfrec_amd_ctx *c;
	bool ok = false;
	explicit PoisonGuard(tfrec_amd_ctx *c_) : c(c_) {}
	~PoisonGuard()
	{
		if (!ok)
			c->poisoned = true;
	}
};
}  // namespace

// ---- biquad coefficients (iir2::set, dsp_stuff.cpp:36-45) in the arithmetic of the reference's normative
// build (oracle/tfrec_oracle.c header).  The five cut-offs the reference ever instantiates
// (main.cpp:186-217, tfa2.cpp:321, whb.cpp:610-611) come from a table of the exact values that build
// produces, so they do not depend on this host's libm tan(); other cut-offs use the formula.
static BiquadCoef biquad_coef(double cutoff)
{
	static const struct {
		double cutoff, b0, a1, a2;
	} known[] = {
		{ 0.5 / (384000.0 / 17240), 0x1.27f98b1037a14p-8, 0x1.cd1527f4a26e2p+0, -0x1.a36a1c41c6995p-1 },
		{ 0.5 / (384000.0 / 9600), 0x1.7ed02b18a270dp-10, 0x1.e397ac010fc89p+0, -0x1.ca2cf85850d62p-1 },
		{ 0.5 / (384000.0 / 8842), 0x1.461fa1a309718p-10, 0x1.e5d4f47377e30p+0, -0x1.ce36282a35d90p-1 },
		{ 2.0 / 64.0, 0x1.14a67102a1ffdp-7, 0x1.b949652fa3970p+0, -0x1.83dd316f714e0p-1 },
		{ 0.0025 / 64.0, 0x1.02ae4cfc8910ap-26, 0x1.ffe9409fe171bp+0, -0x1.ffd283451f7d3p-1 },
	};
	BiquadCoef c;
	for (const auto &k : known)
		if (k.cutoff == cutoff) {
			c.b0 = k.b0;
			c.b1 = k.b0 + k.b0;
			c.b2 = k.b0;
			c.a1 = k.a1;
			c.a2 = k.a2;
			return c;
		}
	const double i = 1.0 / tan(cutoff * M_PI);
	const double s = sqrt(2.0);
	const double b0 = 1.0 / ((i + s) * i + 1.0);
	const double t = i * i - 1.0;
	c.b0 = b0;
	c.b1 = b0 + b0;
	c.b2 = b0;
	c.a1 = (t + t) * b0;
	c.a2 = ((s - i) * i - 1.0) * b0;
	return c;
}

static int d2i_host(double v)
{
	if (!(v > -2147483649.0 && v < 2147483648.0))
		return (int)0x80000000;
	return (int)v;
}

// The discriminator samples the device decided with its exact slow path (fm_resolve.h), checked against the libm of
// THIS host -- the arithmetic the reference binary would use here (dsp_stuff.cpp:284-292 as compiled: DESIGN.md 1).
static void account_fm_log(FmTotals *t, const EventBuf &eb)
{
	t->resolved += eb.uncertain;
	t->undecidable += eb.fm_undecidable;
	const uint32_t n = std::min<uint32_t>(eb.fm_logged, (uint32_t)kFmLogCap);
	const double scale = 16384.0 * (1.0 / M_PI);
	for (uint32_t k = 0; k < n; k++) {
		const int want = d2i_host(atan2(eb.fm_log[k].cj, eb.fm_log[k].cr) * scale);
		t->verified++;
		if (want != eb.fm_log[k].result)
			t->mismatch++;
	}
}

extern "C" {

const char *tfrec_amd_version(void) { return "tfrec_amd 0.1 (gfx950)"; }

int tfrec_amd_fifo_depth(void) { return kSets; }

const char *tfrec_amd_strerror(int code)
{
	switch (code) {
	case TFREC_AMD_OK: return "ok";
	case TFREC_AMD_E_INVAL: return "invalid argument or unsupported configuration";
	case TFREC_AMD_E_NOMEM: return "out of memory";
	case TFREC_AMD_E_HIP: return "HIP runtime error";
	case TFREC_AMD_E_OVERFLOW: return "event buffer overflow";
	case TFREC_AMD_E_STATE: return "call sequence error";
	default: return "unknown error";
	}
}

const char *tfrec_amd_last_error(void) { return g_err; }

int tfrec_amd_rssi_db(int slot, int64_t rssi_raw)
{
	if (slot == TFREC_AMD_SLOT_WHB)  // whb.cpp:696 as compiled: 10*log10(rssi*0.00025 + 1)
		return d2i_host(10 * log10((double)rssi_raw * 0.00025 + 1.0));
	// tfa1.cpp:180, tfa2.cpp:434: (int)(10*log10(rssi)) with an int rssi
	return d2i_host(10 * log10((double)(int)rssi_raw));
}

int tfrec_amd_destroy(tfrec_amd_ctx *c)
{
	if (!c)
		return TFREC_AMD_OK;
	if (TFREC_KNOB_STR("HOST_PROF") && c->hp_n)
		fprintf(stderr, "tfrec_amd host time per batch: submit %.0f us, drain: wait %.0f + copy %.0f + sort %.0f us (%ld batches)\n",
			1e6 * c->hp_submit / c->hp_n, 1e6 * c->hp_wait / c->hp_n, 1e6 * c->hp_copy / c->hp_n, 1e6 * c->hp_sort / c->hp_n,
			c->hp_n);
	if (TFREC_KNOB_STR("HOST_PROF") && c->hp_gap_n)
		fprintf(stderr, "tfrec_amd front-end stream: %.3f ms between one submit's front end and the next one's; front-end start -> TFA_1 chain end %.2f ms; front-end start to start %.3f ms (%ld batches)\n",
			c->hp_gap / c->hp_gap_n, c->hp_lat / c->hp_gap_n, c->hp_s2s / c->hp_gap_n, c->hp_gap_n);
	(void)hipSetDevice(c->cfg.device);
	(void)hipDeviceSynchronize();
	for (int a = 0; a < kNSlots; a++)
		if (c->launch.states[a])
			(void)hipFree(c->launch.states[a]);
	for (int k = 0; k < kSets; k++) {
		(void)hipFree(c->d_dec[k]);
		(void)hipFree(c->d_mask[k]);
		(void)hipFree(c->d_fmdev[k]);
		(void)hipFree(c->d_prevdec[k]);
		if (c->ev_in[k])
			(void)hipEventDestroy(c->ev_in[k]);
		if (c->ev_front[k])
			(void)hipEventDestroy(c->ev_front[k]);
	}
	if (c->fs)
		(void)hipStreamDestroy(c->fs);
	if (c->cp)
		(void)hipStreamDestroy(c->cp);
	if (c->cs)
		(void)hipStreamDestroy(c->cs);
	for (int k = 0; k < kSets; k++) {
		(void)hipFree(c->d_ld16[k]);
		(void)hipFree(c->d_dev32[k]);
		(void)hipFree(c->win_block[k]);
		for (auto &e : c->ev_pipe[k])
			if (e)
				(void)hipEventDestroy(e);
	}
	(void)hipFree(c->d_tcarry);
	if (c->cz)
		(void)hipStreamDestroy(c->cz);
	if (c->fq)
		(void)hipStreamDestroy(c->fq);
	if (c->ks)
		(void)hipStreamDestroy(c->ks);
	if (c->cq)
		(void)hipStreamDestroy(c->cq);
	(void)hipFree(c->d_whbx);
	(void)hipFree(c->d_whbcarry);
	(void)hipFree(c->d_whbgen);
	(void)hipFree(c->d_whbX);
	(void)hipFree(c->d_whbscr);
	for (auto &e : c->ev_aux)
		if (e)
			(void)hipEventDestroy(e);
	if (c->deep && c->k2)
		(void)hipStreamDestroy(c->k2);
	if (c->deep && c->kw)
		(void)hipStreamDestroy(c->kw);
	(void)hipFree(c->d_fsk);
	(void)hipFree(c->d_tail[0]);
	(void)hipFree(c->d_tail[1]);
	for (int k = 0; k < kSets; k++) {
		(void)hipFree(c->d_in16[k]);
		(void)hipFree(c->d_tail10[k]);
	}
	for (int k = 0; k < kSets; k++) {
		(void)hipFree(c->d_evblock[k]);
		for (auto &e : c->done[k])
			if (e)
				(void)hipEventDestroy(e);
		for (auto &e : c->ev[k])
			if (e)
				(void)hipEventDestroy(e);
		for (auto &e : c->tev[k])
			if (e)
				(void)hipEventDestroy(e);
	}
	(void)hipFree(c->d_eb_fresh);
	for (int k = 0; k < kSets; k++) {
		if (c->h_evblock[k])
			(void)hipHostFree(c->h_evblock[k]);
		if (c->copied[k])
			(void)hipEventDestroy(c->copied[k]);
	}
	for (auto &p : c->d_stage)
		(void)hipFree(p);
	if (c->aux)
		(void)hipStreamDestroy(c->aux);
	if (c->t1)
		(void)hipStreamDestroy(c->t1);
	delete c;
	return TFREC_AMD_OK;
}

int tfrec_amd_create(const tfrec_amd_config *cfg, tfrec_amd_ctx **out)
{
	if (!cfg || !out)
		return TFREC_AMD_E_INVAL;
	*out = nullptr;
	if (cfg->n_streams < 1 || cfg->n_streams > 65535 || cfg->max_blocks < 1 || cfg->max_blocks > 4096 ||
	    cfg->max_events < 1 || (cfg->types_mask & 0x2f) == 0 || (cfg->types_mask & ~0x2f) != 0 ||
	    cfg->filter_type < 0 || cfg->filter_type > 1) {
		snprintf(g_err, sizeof(g_err), "bad config");
		return TFREC_AMD_E_INVAL;
	}
	if (cfg->thresh < 0) {
		snprintf(g_err, sizeof(g_err), "thresh must be >= 0 (0 = the reference's auto mode)");
		return TFREC_AMD_E_INVAL;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		snprintf(g_err, sizeof(g_err), "no HIP device available");
		return TFREC_AMD_E_HIP;
	}
	if (cfg->device < 0 || cfg->device >= ndev)
		return TFREC_AMD_E_INVAL;
	HIPCHK(hipSetDevice(cfg->device));

	tfrec_amd_ctx *c = new (std::nothrow) tfrec_amd_ctx();
	if (!c)
		return TFREC_AMD_E_NOMEM;
	c->cfg = *cfg;
	if (const char *fe = TFREC_KNOB_STR("FM_FLAG_EPS"))
		c->fm_flag_eps = std::max(1e-9, atof(fe));
	if (const char *cg = TFREC_KNOB_STR("COPY_GUESS_MIN"))
		c->copy_guess = c->copy_guess_min = (uint32_t)std::max(1, atoi(cg));
	memset(&c->launch, 0, sizeof(c->launch));
	memset(&c->win, 0, sizeof(c->win));

	// second-stage taps: dsp_stuff.cpp:61-88 (narrow) / :91-117 (wide, -W), as h / 65536
	static const int16_t narrow[20] = { -1087, -1082, -1065, -451, 912, 2997, 5556, 8157, 10285, 11484,
					    11484, 10285, 8157, 5556, 2997, 912, -451, -1065, -1082, -1087 };
	static const int16_t wide[20] = { 546, 451, -317, -1844, -3198, -2817, 494, 6469, 13074, 17421,
					  17421, 13074, 6469, 494, -2817, -3198, -1844, -317, 451, 546 };
	for (int n = 0; n < 20; n++)
		c->taps.f2[n][0] = c->taps.f2[n][1] = (float)(cfg->filter_type ? wide[n] : narrow[n]) / 65536.0f;

	// registration order, types and samples-per-bit of main.cpp:173-218
	static const struct {
		int sensor_type, kind, min_bytes;
		double baud;
	} reg[kNSlots] = { { 0, 0, 10, 0 }, { 1, 1, 7, 17240 }, { 2, 1, 7, 9600 }, { 3, 1, 7, 8842 }, { 5, 2, 11, 6000 } };
	const size_t m_max = (size_t)cfg->max_blocks * kBlockDec;
	const size_t n = (size_t)cfg->n_streams;
	int rc = TFREC_AMD_OK;
#define ALLOC(ptr, bytes)                                                     \
	do {                                                                  \
		if (rc == TFREC_AMD_OK && hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) { \
			snprintf(g_err, sizeof(g_err), "hipMalloc(%zu) failed", (size_t)(bytes)); \
			rc = TFREC_AMD_E_NOMEM;                               \
		} else if (rc == TFREC_AMD_OK) {                              \
			c->dev_bytes += (size_t)(bytes);                      \
		}                                                             \
	} while (0)
	for (int s = 0; s < kNSlots && rc == TFREC_AMD_OK; s++) {
		if (!(cfg->types_mask & (1 << reg[s].sensor_type)))
			continue;
		const int a = c->launch.n_active++;
		c->launch.slot[a] = s;
		ChainParams &p = c->launch.params[a];
		memset(&p, 0, sizeof(p));
		p.kind = reg[s].kind;
		p.sensor_type = reg[s].sensor_type;
		p.min_bytes = reg[s].min_bytes;
		if (p.kind == 0) {
			p.window = 400;  // 40*BITPERIOD, tfa1.cpp:34, 148
			p.spb = 10;
		} else {
			p.spb = (1536000 / 4.0) / reg[s].baud;
			if (p.kind == 1) {
				p.window = d2i_host(16 * p.spb);  // tfa2.cpp:355
				// multiplier form of numbits (ChainParams::nb_mul), accepted only if it reproduces the fp64 expression
				// for every argument the slicer can form
				const int hmax = (int)(16 * p.spb) + 2;
				const uint64_t a0 = (uint64_t)((double)(1ull << 40) / p.spb);
				for (uint64_t a_try : { a0, a0 + 1, a0 - 1 }) {
					bool ok = true;
					for (int h = 0; h <= hmax && ok; h++)
						ok = tfa2_numbits_mul(2 * h, a_try) == (int)(((double)h + p.spb / 2) / p.spb);
					if (ok) {
						p.nb_mul = a_try;
						break;
					}
				}
				p.td_lo = (int)floor(p.spb / 4) + 1;
				p.td_hi = (int)ceil(32 * p.spb) - 1;
				p.iir = biquad_coef(0.5 / p.spb);  // tfa2.cpp:321
			} else {
				// whb_demod_kernel's candidate walk has the reference's 64 samples per bit (main.cpp:217) built in (chains2.hip: kWhbSpb)
				if (p.spb != 64.0) {
					snprintf(g_err, sizeof(g_err), "WHB chain with %.3f samples per bit: only 64 is built", p.spb);
					rc = TFREC_AMD_E_INVAL;
					break;
				}
				p.window = d2i_host(8 * p.spb);         // whb.cpp:641
				p.iir = biquad_coef(2.0 / p.spb);       // whb.cpp:610
				p.iir_avg = biquad_coef(0.0025 / p.spb); // whb.cpp:611
			}
		}
		ALLOC(c->launch.states[a], n * sizeof(ChainState));
		if (rc != TFREC_AMD_OK)
			break;
		// constructor state: tfa1.cpp:136-141, tfa2.cpp:316-334, whb.cpp:605-623, decoders :36-45/:54-62/:77-107
		std::vector<ChainState> init(n);
		memset(init.data(), 0, n * sizeof(ChainState));
		for (auto &st : init) {
			st.sr_cnt = -1;
			st.dmin = 32767;
			st.dmax = -32767;
		}
		if (hipMemcpy(c->launch.states[a], init.data(), n * sizeof(ChainState), hipMemcpyHostToDevice) != hipSuccess)
			rc = TFREC_AMD_E_HIP;
	}
	c->dec_stride = m_max;
	c->mask_stride = m_max / 64;
	for (int k = 0; k < kSets; k++) {
		ALLOC(c->d_dec[k], n * c->dec_stride * sizeof(uint32_t) + 256);  // + slack: K3 loads whole 32-sample chunks at window tails
		ALLOC(c->d_mask[k], n * c->mask_stride * sizeof(unsigned long long));
	}
	for (int a = 0; a < c->launch.n_active; a++)
		c->wmax = std::max(c->wmax, (int)c->launch.params[a].window);
	if (cfg->thresh == 0) {  // fm_demod.cpp:23-27: 0 selects the adaptive mode starting at 500
		ALLOC(c->d_fsk, n * sizeof(FskState));
		if (rc == TFREC_AMD_OK) {
			std::vector<FskState> init(n, FskState{ 500, 0, 0, -(1 << 28) });
			if (hipMemcpy(c->d_fsk, init.data(), n * sizeof(FskState), hipMemcpyHostToDevice) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
	}
	for (int k = 0; k < kSets; k++) {
		ALLOC(c->d_fmdev[k], n * m_max * sizeof(int16_t) + 256);  // + slack: K3 reads whole dwords past an odd tail
		ALLOC(c->d_prevdec[k], n * sizeof(uint32_t));
	}
	for (int a = 0; a < c->launch.n_active; a++)
		c->need_fmdev = c->need_fmdev || c->launch.params[a].kind == 1;
	if (!(cfg->flags & TFREC_AMD_F_SERIAL_CHAINS)) {
		// window-parallel pipeline buffers (chains2.hip)
		const size_t chains = (size_t)c->launch.n_active * n;
		bool whb = false;
		for (int a = 0; a < c->launch.n_active; a++)
			whb = whb || c->launch.params[a].kind == 2;
		ALLOC(c->d_tcarry, chains * 4);
		if (rc == TFREC_AMD_OK && hipMemset(c->d_tcarry, 0, chains * 4) != hipSuccess)
			rc = TFREC_AMD_E_HIP;
		if (whb) {  // iir_avg starts from zero like every iir2 (dsp_stuff.cpp:28-34)
			ALLOC(c->d_whbx, n * sizeof(WhbExact));
			ALLOC(c->d_whbcarry, n * sizeof(int));
			ALLOC(c->d_whbgen, n * sizeof(uint32_t));
			ALLOC(c->d_whbX, n * sizeof(ChainState));
			ALLOC(c->d_whbscr, n * sizeof(ChainState));
			// TEST hooks (results stay exact under both: the check's tolerance and the ambiguity rule widen with D, and a forced
			// failure is only a redo) -- clamped, and never silent: a stray variable changes the redo rate, i.e. the speed
			if (const char *tp = TFREC_KNOB_STR("WHB_TEST_PERTURB"))
				c->whb_test_perturb = std::max(-1000000, std::min(1000000, atoi(tp)));
			if (rc == TFREC_AMD_OK && hipMemset(c->d_whbgen, 0, n * sizeof(uint32_t)) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
			if (const char *ff = TFREC_KNOB_STR("WHB_FORCE_FAIL"))
				c->whb_force_fail = std::max(0, atoi(ff));
#if TFREC_KNOBS_BUILT
			if (c->whb_test_perturb || c->whb_force_fail)
				fprintf(stderr, "tfrec_amd: TEST hook active (TFREC_AMD_WHB_TEST_PERTURB=%d, TFREC_AMD_WHB_FORCE_FAIL=%d): WHB streams are "
						"redone on purpose, results unchanged, throughput lower\n", c->whb_test_perturb, c->whb_force_fail);
#endif
			if (rc == TFREC_AMD_OK && (hipMemset(c->d_whbx, 0, n * sizeof(WhbExact)) != hipSuccess ||
						   hipMemset(c->d_whbcarry, 0, n * sizeof(int)) != hipSuccess))
				rc = TFREC_AMD_E_HIP;
		}
		for (int set = 0; set < kSets; set++) {
		WinTables &T = c->win[set];
		T.cap = (int32_t)(m_max / 356 + 2);  // windows of one chain are > W-1 >= 355 samples apart
		T.slots = (int32_t)(m_max / 32 + (size_t)T.cap + 2);  // window-relative 32-sample slots per chain row
		if ((size_t)T.slots > (size_t)kWhbRecOffMask) {  // WhbStepRec::meta packs a slot index into kWhbRecOffMask's bits
			snprintf(g_err, sizeof(g_err), "max_blocks too large for the WHB step records");
			rc = TFREC_AMD_E_INVAL;
			break;
		}
		// rows only for the chains that use them: ld16 for the TFA_2 family (its slots are adjacent in registration order),
		// checkpoints for the chains with a biquad stage (all but TFA_1, which is registered first)
		int a_ld0 = -1, n_ld = 0, a_ck0 = -1;
		for (int a = 0; a < c->launch.n_active; a++) {
			if (c->launch.params[a].kind == 1) {
				if (a_ld0 < 0)
					a_ld0 = a;
				n_ld = a - a_ld0 + 1;
			}
			if (c->launch.params[a].kind != 0 && a_ck0 < 0)
				a_ck0 = a;
		}
		T.ld_c0 = (int32_t)((a_ld0 < 0 ? 0 : a_ld0) * n);
		T.ck_c0 = (int32_t)((a_ck0 < 0 ? 0 : a_ck0) * n);
		const size_t ck_chains = a_ck0 < 0 ? 0 : chains - (size_t)a_ck0 * n;
		ALLOC(c->d_ld16[set], std::max<size_t>(1, (size_t)n_ld * n) * (size_t)T.slots * 32 * sizeof(int16_t));
		if (whb)
			ALLOC(c->d_dev32[set], n * (size_t)T.slots * 32 * sizeof(int32_t) + 4096);  // + slack: whb_demod_kernel keeps two 64-sample steps in flight past a row's last window
		T.bit_words = (int32_t)(m_max / 64 + 3 * (size_t)T.cap + 8);
		const size_t wins = chains * (size_t)T.cap;
		size_t off = 0;
		auto carve = [&](size_t bytes) {
			const size_t o = off;
			off += (bytes + 255) & ~(size_t)255;
			return o;
		};
		const size_t o_count = carve(chains * 4), o_cont = carve(chains * 4), o_tnext = carve(chains * 4);
		const size_t o_open = carve(wins * 4), o_close = carve(wins * 4), o_res = carve(wins * sizeof(WinResult));
		const size_t o_dcd = carve(wins * sizeof(WinDecode)), o_wst = carve(whb ? n * (size_t)T.cap * sizeof(WhbStart) : 0);
		const size_t o_bits = carve(chains * (size_t)T.bit_words * 4), o_items = carve((kNQueues * wins + chains) * sizeof(uint2));
		const size_t o_queue = carve((kNQueues + 1) * sizeof(WorkQueue)), o_stats = carve(128);
		T.segcap = (int32_t)((m_max / 32 + (size_t)T.cap) / kSegSlots + 2);
		const size_t segs = chains * (size_t)T.segcap;
		const size_t o_ckpt = carve(ck_chains * (size_t)T.slots * sizeof(double2));
		const size_t o_sstart = carve(segs * sizeof(uint2)), o_vtotal = carve(chains * 4);
		const size_t o_se1 = carve(segs * sizeof(BiquadEnd)), o_se2 = carve(segs * sizeof(BiquadEnd)), o_sfix = carve(segs * 4);
		const size_t o_se3 = carve(segs * sizeof(BiquadEnd)), o_sfix2 = carve(segs * 4);
		const size_t o_cand = carve(n * (size_t)T.slots * 4), o_mark = carve(n * (size_t)T.slots * sizeof(MarkPiece));
		T.whbrec_stride = (int32_t)(m_max / 64 + 2 * (size_t)T.cap + 2 + kWhbRecSlack);
		const size_t o_wrec = carve(whb ? n * (size_t)T.whbrec_stride * sizeof(WhbStepRec) : 0), o_wfail = carve(whb ? n * 4 : 0);
		const size_t o_wsnap = carve(whb ? n * sizeof(ChainState) : 0), o_wx0 = carve(whb ? n * sizeof(WhbExact) : 0);
		const size_t o_wseen = carve(whb ? n * 4 : 0);
		T.whbdense_stride = (int32_t)(m_max + 64);
		T.whbx_stride = (int32_t)(m_max / 64 + 2);
		const size_t o_wdn = carve(whb ? n * 4 : 0), o_wxb = carve(whb ? n * (size_t)T.whbx_stride * 8 : 0);
		const size_t o_wxs = carve(whb ? n * (size_t)T.whbx_stride * sizeof(double2) : 0);
		const size_t o_wdense = carve(whb ? n * (size_t)T.whbdense_stride * 4 : 0);
		ALLOC(c->win_block[set], off);
		if (rc == TFREC_AMD_OK) {
			uint8_t *b = (uint8_t *)c->win_block[set];
			T.count = (int32_t *)(b + o_count);
			T.cont = (int32_t *)(b + o_cont);
			T.timeout_next = (int32_t *)(b + o_tnext);
			T.open = (int32_t *)(b + o_open);
			T.close = (int32_t *)(b + o_close);
			T.result = (WinResult *)(b + o_res);
			T.decode = (WinDecode *)(b + o_dcd);
			T.whbstart = (WhbStart *)(b + o_wst);
			T.bits = (uint32_t *)(b + o_bits);
			T.items = (uint2 *)(b + o_items);
			T.queue = (WorkQueue *)(b + o_queue);
			T.overflow = nullptr;  // lives in the set's event block (kEvOverflowOff): set below, reset by every submit
			T.stats = (unsigned long long *)(b + o_stats);
			T.ckpt = (double2 *)(b + o_ckpt);
			T.segstart = (uint2 *)(b + o_sstart);
			T.vtotal = (int32_t *)(b + o_vtotal);
			T.segend1 = (BiquadEnd *)(b + o_se1);
			T.segend2 = (BiquadEnd *)(b + o_se2);
			T.segfix = (int32_t *)(b + o_sfix);
			T.segend3 = (BiquadEnd *)(b + o_se3);
			T.segfix2 = (int32_t *)(b + o_sfix2);
			T.cand = (uint32_t *)(b + o_cand);
			T.mark = (MarkPiece *)(b + o_mark);
			T.whbrec = (WhbStepRec *)(b + o_wrec);
			T.whbfail = (int32_t *)(b + o_wfail);
			T.whbsnap = (ChainState *)(b + o_wsnap);
			T.whbx0 = (WhbExact *)(b + o_wx0);
			T.whbseen = (uint32_t *)(b + o_wseen);
			T.whbdense = (int32_t *)(b + o_wdense);
			T.whbdense_n = (int32_t *)(b + o_wdn);
			T.whbxbits = (unsigned long long *)(b + o_wxb);
			T.whbxsnap = (double2 *)(b + o_wxs);
			T.whbgen = c->d_whbgen;
			T.whbX = c->d_whbX;
			T.whbscr = c->d_whbscr;
			T.whbpub = nullptr;
			T.whb_test_perturb = c->whb_test_perturb;
			T.tfa1_vec = TFREC_KNOB_INT("TFA1_VEC", 1, 0, 1) != 0;
			T.tfa2_vec = TFREC_KNOB_INT("TFA2_VEC", 1, 0, 1) != 0;
			T.whb_force_fail = c->whb_force_fail;
			T.whbx = c->d_whbx;
			T.timeout_carry = c->d_tcarry;
			T.prevdec = c->d_prevdec[set];
			if (hipMemset(T.queue, 0, (kNQueues + 1) * sizeof(WorkQueue)) != hipSuccess ||
			    hipMemset(T.stats, 0, 128) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
		}
	}
	c->in10x = (cfg->flags & TFREC_AMD_F_INPUT_10X) != 0;
	const size_t tail_bytes = c->in10x ? 2 * (size_t)kTailBytes : (size_t)kTailBytes;  // int16 history is twice as wide
	ALLOC(c->d_tail[0], n * tail_bytes);
	ALLOC(c->d_tail[1], n * tail_bytes);
	if (c->in10x) {
		c->in16_stride = 4 * m_max;  // complex samples at 1.536 MS/s per stream and submit
		for (int k = 0; k < kSets; k++) {
			ALLOC(c->d_in16[k], n * c->in16_stride * sizeof(uint32_t));
			ALLOC(c->d_tail10[k], n * 112);
		}
	}
	// One block per set: [EventBuf | the window tables' overflow flag, 16 B | pad to 256 | events]: the drain's device-to-host
	// copy of a submit is ONE copy on the stream that sets the batch period (three copies were 0.45 ms of it with their gaps)
	for (int k = 0; k < kSets; k++) {
		ALLOC(c->d_evblock[k], kEvHeader + (size_t)cfg->max_events * sizeof(tfrec_amd_event));
		if (rc == TFREC_AMD_OK) {
			c->d_eb[k] = (EventBuf *)c->d_evblock[k];
			c->d_events[k] = (tfrec_amd_event *)(c->d_evblock[k] + kEvHeader);
			if (c->win[k].count)  // (window-parallel pipeline: its overflow flag lives behind the EventBuf)
				c->win[k].overflow = (int32_t *)(c->d_evblock[k] + kEvOverflowOff);
		}
	}
	ALLOC(c->d_eb_fresh, kEvFreshBytes);
#undef ALLOC
	for (int k = 0; k < kSets && rc == TFREC_AMD_OK; k++) {
		if (hipHostMalloc((void **)&c->h_evblock[k], kEvHeader + (size_t)cfg->max_events * sizeof(tfrec_amd_event), hipHostMallocDefault) != hipSuccess ||
		    hipHostGetDevicePointer((void **)&c->h_evblock_dev[k], c->h_evblock[k], 0) != hipSuccess ||
		    hipEventCreateWithFlags(&c->copied[k], hipEventDisableTiming) != hipSuccess) {
			rc = TFREC_AMD_E_NOMEM;
			break;
		}
		c->pinned_bytes += kEvHeader + (size_t)cfg->max_events * sizeof(tfrec_amd_event);
		c->h_eb[k] = (EventBuf *)c->h_evblock[k];
		c->h_events[k] = (tfrec_amd_event *)(c->h_evblock[k] + kEvHeader);
	}
	// Every pipeline stream except the biquad stages runs at high priority.  With the front end at low priority
	// ("fill what the latency-bound chains leave free") its kernel stretched from 3 to 11 ms beside the chains and,
	// with three submits in flight, became the longest stage of all: 13.4 ms per batch instead of 11.7.
	int prio_lo = 0, prio_hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	const int prio_fs = TFREC_KNOB_STR("PRIO_FS") ? atoi(TFREC_KNOB_STR("PRIO_FS")) : prio_hi;
	// (experiments: TFREC_AMD_PRIO = one letter h / n / l per stream in the order fs cp cs t1 aux k2 kw, default "hnhhhnn")
	const char *prio_env = TFREC_KNOB_STR("PRIO");
	auto prio_of = [&](int k, int dflt) {
		if (!prio_env || strlen(prio_env) <= (size_t)k)
			return dflt;
		return prio_env[k] == 'h' ? prio_hi : (prio_env[k] == 'l' ? prio_lo : 0);
	};
	auto mkstream = [&](hipStream_t *st, int k, int dflt) { return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_of(k, dflt)); };
	// (experiment, TFREC_AMD_FS_CUS=n: the front end's stream confined to n of the chip's compute units (a CU mask; such a stream has
	// normal priority) -- at high priority its 196 k workgroups hold the workgroup dispatcher for their whole duration and nothing else
	// STARTS meanwhile: profiles/r06_final_steps.txt)
	auto mk_fs = [&]() -> hipError_t {
		const int ncu = TFREC_KNOB_INT("FS_CUS", 0, 0, 256);
		if (ncu <= 0)
			return mkstream(&c->fs, 0, prio_fs);
		uint32_t mask[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		const int stride = TFREC_KNOB_INT("FS_CU_STRIDE", 1, 1, 8);  // 1: the lowest n bits; k: every k-th bit
		for (int i = 0, set = 0; i < 256 && set < ncu; i += stride, set++)
			mask[i >> 5] |= 1u << (i & 31);
		return hipExtStreamCreateWithCUMask(&c->fs, 8, mask);
	};
	if (rc == TFREC_AMD_OK) {
		// zero FIR history == u8 value 128 (decimate::decimate zeroes hist0, dsp_stuff.cpp:145-152)
		EventBuf eb;
		memset(&eb, 0, sizeof(eb));
		eb.capacity = (uint32_t)cfg->max_events;
		// (int16 history of the 10x path: zero; raw u8 history of its 10:1 stage: 128)
		if (hipMemset(c->d_tail[0], c->in10x ? 0 : 0x80, n * tail_bytes) != hipSuccess ||
		    hipMemset(c->d_tail[1], c->in10x ? 0 : 0x80, n * tail_bytes) != hipSuccess ||
		    (c->in10x && (hipMemset(c->d_tail10[0], 0x80, n * 112) != hipSuccess ||
				  hipMemset(c->d_tail10[1], 0x80, n * 112) != hipSuccess)) ||
		    hipMemset(c->d_eb_fresh, 0, kEvFreshBytes) != hipSuccess ||
		    hipMemcpy(c->d_eb_fresh, &eb, sizeof(eb), hipMemcpyHostToDevice) != hipSuccess ||
		    mk_fs() != hipSuccess ||
		    mkstream(&c->cp, 1, 0) != hipSuccess ||
		    mkstream(&c->cs, 2, prio_hi) != hipSuccess ||
		    false)
			rc = TFREC_AMD_E_HIP;
		for (int k = 0; k < kSets && rc == TFREC_AMD_OK; k++) {
			for (auto &e : c->done[k])
				if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
					rc = TFREC_AMD_E_HIP;
			for (auto &e : c->ev_pipe[k])
				if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
					rc = TFREC_AMD_E_HIP;
			if (hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&c->ev_front[k], hipEventDisableTiming) != hipSuccess ||
			    hipMemset(c->d_evblock[k], 0, kEvHeader) != hipSuccess ||
			    hipMemcpy(c->d_eb[k], &eb, sizeof(eb), hipMemcpyHostToDevice) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
	}
	if (rc == TFREC_AMD_OK && !(cfg->flags & TFREC_AMD_F_SERIAL_CHAINS)) {
		if (mkstream(&c->t1, 3, prio_hi) != hipSuccess || mkstream(&c->aux, 4, prio_hi) != hipSuccess)
			rc = TFREC_AMD_E_HIP;
		// deep layout (default; TFREC_AMD_DEEP=0 selects the shallow one): the biquad stages get streams of their own
		const char *dp = TFREC_KNOB_STR("DEEP");
		c->deep = dp ? atoi(dp) != 0 : true;
		// The discriminator pass moves from the front-end stream to the head of the TFA_2-family biquad stage when a WHB
		// demodulator is registered: then the WHB chain is the longest and the front-end stream the busiest (measured
		// 12.2 -> 11.8 ms per batch; without WHB the TFA_2 chain is the longest and the move costs 7.7 -> 8.5 ms)
		bool has_whb = false;
		for (int a = 0; a < c->launch.n_active; a++)
			has_whb = has_whb || c->launch.params[a].kind == 2;
		c->scan_on_kw = TFREC_KNOB_INT("SCAN_KW", 1, 0, 1) != 0;
		c->fmdev_k2 = TFREC_KNOB_INT("FMDEV_K2", has_whb ? 1 : 0, 0, 1) != 0;
		c->k2 = c->cs;
		c->kw = c->aux;
		c->vx = c->aux;
		if (c->deep && rc == TFREC_AMD_OK &&
		    (mkstream(&c->k2, 5, 0) != hipSuccess || mkstream(&c->kw, 6, 0) != hipSuccess))
			rc = TFREC_AMD_E_HIP;
		// TFREC_AMD_COOP_STREAM=1: a stream for the TFA_2 family's cooperative slicers (PipeCtl::cz).  It is the fifth of high
		// priority: with the HIP default of four hardware queues per priority it shares one (GPU_MAX_HW_QUEUES >= 8 wanted).
		if (c->deep && rc == TFREC_AMD_OK && TFREC_KNOB_INT("COOP_STREAM", 0, 0, 2) != 0 &&
		    hipStreamCreateWithPriority(&c->cz, hipStreamNonBlocking, TFREC_KNOB_INT("COOP_STREAM", 0, 0, 2) == 2 ? 0 : prio_hi) != hipSuccess)
			rc = TFREC_AMD_E_HIP;
		// The discriminator pass on a LOW-priority stream of its own (TFREC_AMD_FMDEV_OWN: 0 = at the head of k2, 1 = low
		// (default when a WHB demodulator is registered), 2 = normal, 3 = high priority).  It needs the front end only, not
		// the window scan; at the head of k2 it made that stream (discriminator + five biquad kernels) the one that set the
		// batch period: 6.98 -> 6.60 ms per batch over 100 steps (profiles/r04_ab_fmdev_stream.txt).  The low-priority pool's
		// hardware queues are otherwise unused, so the stream shares none (a fifth normal-priority stream would).
		{
			const int m = TFREC_KNOB_INT("FMDEV_OWN", c->fmdev_k2 ? 1 : 0, 0, 3);
			if (c->deep && rc == TFREC_AMD_OK && m > 0 && c->need_fmdev && c->fmdev_k2 &&
			    hipStreamCreateWithPriority(&c->fq, hipStreamNonBlocking, m == 1 ? prio_lo : (m == 2 ? 0 : prio_hi)) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
		// The speculative biquad pass of the TFA_2 family on a stream of its own (TFREC_AMD_SPEC_OWN: 0 = at the head of k2,
		// 1 = low priority (default when the discriminator pass has its stream), 2 = normal, 3 = high): it needs nothing of the
		// submit before (chains2.hip K3a), so it runs beside that submit's repair passes and chain walk on k2.  The third
		// stream of the low-priority pool (with fq and cq): no shared hardware queue.
		{
			const int m = TFREC_KNOB_INT("SPEC_OWN", 1, 0, 3);
			if (c->deep && rc == TFREC_AMD_OK && m > 0 && c->fq &&
			    hipStreamCreateWithPriority(&c->ks, hipStreamNonBlocking, m == 1 ? prio_lo : (m == 2 ? 0 : prio_hi)) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
		// The drain's copy on a low-priority stream of its own (TFREC_AMD_COPY_OWN=0: on cp).  On cp it sat between the WHB
		// checks of consecutive submits and waits for ALL chains of its submit: the check of submit k + 1 could not start
		// before the TFA chains of submit k had ended.  6.32 -> 6.21 ms per batch over 100 steps
		// (profiles/r04_ab_copy_stream.txt); the low pool's hardware queues hold only this stream and the discriminator's.
		{  // (TFREC_AMD_COPY_OWN: 0 = on cp, 1 = low priority (default), 2 = normal, 3 = high)
			const int m = TFREC_KNOB_INT("COPY_OWN", 1, 0, 3);
			if (c->deep && rc == TFREC_AMD_OK && m > 0 &&
			    hipStreamCreateWithPriority(&c->cq, hipStreamNonBlocking, m == 1 ? prio_lo : (m == 2 ? 0 : prio_hi)) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
		}
		// whb_verify_kernel runs on the copy stream, ahead of its submit's device-to-host copies (they wait for it anyway).
		// A stream of its own would be the FIFTH of normal priority in the process (k2, kw, cp and the caller's): it shared a
		// hardware queue with kw, and the 6 ms verification of submit k held up the WHB biquads of submit k + 2.
		if (c->deep)
			c->vx = c->cp;
		for (int k = 0; k < kSets && rc == TFREC_AMD_OK; k++)
			if (hipEventCreateWithFlags(&c->ev_aux[k], hipEventDisableTiming) != hipSuccess)
				rc = TFREC_AMD_E_HIP;
	}
	if (rc == TFREC_AMD_OK && (cfg->flags & TFREC_AMD_F_TIMING)) {
		for (int k = 0; k < kSets; k++) {
			for (auto &e : c->ev[k])
				if (hipEventCreate(&e) != hipSuccess)
					rc = TFREC_AMD_E_HIP;
			if (!(cfg->flags & TFREC_AMD_F_SERIAL_CHAINS))
				for (auto &e : c->tev[k])
					if (hipEventCreate(&e) != hipSuccess)
						rc = TFREC_AMD_E_HIP;
		}
	}
	for (int a = 0; a < c->launch.n_active; a++)
		c->whb_active = c->whb_active || c->launch.params[a].kind == 2;
	if (rc != TFREC_AMD_OK) {
		tfrec_amd_destroy(c);
		return rc;
	}
	*out = c;
	return TFREC_AMD_OK;
}

// in_stream_is_fs: the input was produced on the front-end stream itself (staged host input): no event needed
static int submit_common(tfrec_amd_ctx *c, const void *d_iq, size_t stride, int n_blocks, void *hip_stream, bool input_on_fs)
{
	if (!c || !d_iq || n_blocks < 1 || n_blocks > c->cfg.max_blocks)
		return TFREC_AMD_E_INVAL;
	const size_t block_bytes = c->in10x ? TFREC_AMD_BLOCK_BYTES_10X : TFREC_AMD_BLOCK_BYTES;
	if ((stride % 16) != 0 || ((uintptr_t)d_iq % 16) != 0 ||
	    (c->cfg.n_streams > 1 && stride < (size_t)n_blocks * block_bytes)) {
		snprintf(g_err, sizeof(g_err), "IQ base and stream stride must be 16-byte aligned and >= one stream");
		return TFREC_AMD_E_INVAL;
	}
	if (c->inflight >= kSets) {
		snprintf(g_err, sizeof(g_err), "%d submits are waiting to be drained: call tfrec_amd_drain_events first", kSets);
		return TFREC_AMD_E_STATE;
	}
	if (c->poisoned) {
		snprintf(g_err, sizeof(g_err), "an earlier submit failed half way: the context must be recreated");
		return TFREC_AMD_E_STATE;
	}
	HIPCHK(hipSetDevice(c->cfg.device));
	PoisonGuard guard(c);  // from here on work is enqueued: a failure leaves the carried state undefined
	const bool timing = (c->cfg.flags & TFREC_AMD_F_TIMING) != 0;
	const int set = (c->head + c->inflight) % kSets;  // this submit's event buffers and timing events
	// Front end on its own stream: it starts when the caller's stream has produced the input, and may overlap the
	// chains of the previous submit (different buffer set; the set's previous user was drained, see the FIFO rule)
	hipStream_t fs = c->fs;
	hipStream_t st = c->cs;  // the chains run on an internal stream: nothing of ours is queued on the caller's
	if (!input_on_fs) {
		HIPCHK(hipEventRecord(c->ev_in[set], (hipStream_t)hip_stream));
		HIPCHK(hipStreamWaitEvent(fs, c->ev_in[set], 0));
	}
	HIPCHK(hipMemcpyAsync(c->d_eb[set], c->d_eb_fresh, kEvFreshBytes, hipMemcpyDeviceToDevice, fs));  // (+ the overflow flag)
	if (timing)
		HIPCHK(hipEventRecord(c->ev[set][0], fs));
	const uint8_t *fin = (const uint8_t *)d_iq;
	size_t fstride = stride;
	if (c->in10x) {  // 15.36 MS/s u8 -> 1.536 MS/s int16 pairs, then the standard cascade on int16 input
		HIPCHK(launch_decim10(fs, (const uint8_t *)d_iq, stride, c->cfg.n_streams, n_blocks, c->d_tail10[c->tail_sel],
				      c->d_tail10[c->tail_sel ^ 1], c->d_in16[set], c->in16_stride));
		fin = (const uint8_t *)c->d_in16[set];
		fstride = c->in16_stride * sizeof(uint32_t);
	}
	// (TFREC_AMD_SKIP bit 512, WHAT-IF timing only: after the first 12 submits the front end is left out and the chains run
	// on what the buffer set holds from four submits ago -- what the front end costs the batch period)
	static const bool whatif_no_fe = (TFREC_KNOB_INT("SKIP", 0, 0, 1 << 16) & 512) != 0;
	static int whatif_submits = 0;
	if (!(whatif_no_fe && ++whatif_submits > 12))
	HIPCHK(launch_frontend(fs, fin, fstride, c->cfg.n_streams, n_blocks, c->d_tail[c->tail_sel],
			       c->d_tail[c->tail_sel ^ 1], c->d_dec[set], c->dec_stride, c->d_mask[set], c->mask_stride,
			       c->d_prevdec[set], c->cfg.thresh ? c->cfg.thresh : 500, c->taps, c->in10x));
	if (c->d_fsk)  // auto threshold: per-block thresholds rewrite the trigger mask (fm_demod.cpp:58-73)
		HIPCHK(launch_threshold(fs, c->d_dec[set], c->dec_stride, c->d_mask[set], c->mask_stride, c->cfg.n_streams,
					n_blocks, c->d_fsk, c->wmax));
	if (timing)
		HIPCHK(hipEventRecord(c->ev[set][3], fs));
	const bool fmdev_k2 = c->need_fmdev && c->fmdev_k2 && !(c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS);
	if (c->need_fmdev && !fmdev_k2)  // FM discriminator of the samples near trigger windows (after the mask is final)
		HIPCHK(launch_fmdev(fs, c->d_dec[set], c->dec_stride, c->d_mask[set], c->mask_stride, c->d_prevdec[set],
				    c->d_fmdev[set], c->dec_stride, c->d_eb[set], c->cfg.n_streams, n_blocks, c->wmax,
				    c->fm_flag_eps));
	if (timing)
		HIPCHK(hipEventRecord(c->ev[set][1], fs));
	HIPCHK(hipEventRecord(c->ev_front[set], fs));
	if (c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS) {
		HIPCHK(hipStreamWaitEvent(st, c->ev_front[set], 0));
		HIPCHK(launch_chains(st, c->d_dec[set], c->dec_stride, c->d_mask[set], c->mask_stride, c->cfg.n_streams, n_blocks,
				     c->sample_base, c->launch, c->d_events[set], c->d_eb[set], c->cfg.flags));
		if (timing)
			HIPCHK(hipEventRecord(c->ev[set][2], st));
		for (auto &e : c->done[set])
			HIPCHK(hipEventRecord(e, st));
	} else {
		PipeCtl P;
		P.fs = fs;
		P.ws = (c->deep && c->scan_on_kw) ? c->kw : fs;
		P.ev_front = c->ev_front[set];
		P.k2 = c->k2;
		P.kw = c->kw;
		P.cs = c->cs;
		P.aux = c->aux;
		P.t1 = c->t1;
		P.vx = c->vx;
		P.ev_aux = c->ev_aux[set];
		P.whb_carry = c->d_whbcarry;
		P.ev_win = c->ev_pipe[set][0];
		P.ev_fork = c->ev_pipe[set][1];
		P.ev_k2 = c->ev_pipe[set][2];
		P.ev_kw = c->ev_pipe[set][3];
		P.ev_fm = c->ev_pipe[set][4];
		P.cz = c->cz;
		P.fq = c->fq;
		P.ks = c->ks;
		P.ev_spec = c->ev_pipe[set][7];
		P.ev_heads = c->ev_pipe[set][5];
		P.ev_coop = c->ev_pipe[set][6];
		for (int k = 0; k < 3; k++)
			P.done[k] = c->done[set][k];
		P.tev = (timing && c->tev[set][0]) ? c->tev[set] : nullptr;
		P.fmdev_wmax = fmdev_k2 ? c->wmax : 0;
		P.fm_flag_eps = c->fm_flag_eps;
		P.fmdev_out = c->d_fmdev[set];
		P.prevdec = c->d_prevdec[set];
		c->win[set].whb_submit_seq = c->submit_seq++;
		HIPCHK(launch_pipeline(P, c->d_dec[set], c->dec_stride, c->d_mask[set], c->mask_stride, c->d_fmdev[set], c->dec_stride,
				       c->cfg.n_streams, n_blocks, c->sample_base, c->launch, c->win[set], c->d_ld16[set],
				       c->d_dev32[set], c->d_events[set], c->d_eb[set], c->cfg.flags));
	}
#ifdef TFREC_AMD_VECSTAT
	if (c->submit_seq == 3) {
		(void)hipDeviceSynchronize();
		unsigned long long st[16] = { 0 };
		(void)hipMemcpy(st, c->win[set].stats, sizeof(st), hipMemcpyDeviceToHost);
		fprintf(stderr, "VECSTAT (one submit) TFA_1: groups %llu, stale piece %llu, entered-with-none hazard %llu, > 64 bits in a lane %llu, lanes with 32 ones or more %llu; TFA_2 family: groups %llu, entered with relative 0 %llu, > 16 rounds %llu, > 64 bits in a lane %llu, walks of the groups that converged %llu\n", st[8], st[9], st[10], st[11], st[5], st[12], st[13], st[14], st[15], st[6]);
	}
#endif
#ifdef TFREC_AMD_COOPSTAT
	if (c->submit_seq == 3) {
		(void)hipDeviceSynchronize();
		unsigned long long st[16] = { 0 };
		(void)hipMemcpy(st, c->win[set].stats, sizeof(st), hipMemcpyDeviceToHost);
		fprintf(stderr, "COOPSTAT (one submit) TFA_2 family: frozen one-block steps %llu, accepted %llu, rejected %llu, other frozen steps %llu; TFA_1: steps %llu, candidate runs %llu; TFA_2 walked steps with a full mask %llu, candidates in walked steps %llu, walked steps that begin inside a run %llu\n", st[7], st[8], st[9], st[10], st[11], st[12], st[13], st[14], st[15]);
	}
#endif
	if (TFREC_KNOB_STR("DEBUG_WINHIST") && c->submit_seq == 3) {  // (debug: the window length distribution of one submit)
		(void)hipDeviceSynchronize();
		const WinTables &T = c->win[set];
		const size_t chains = (size_t)c->launch.n_active * c->cfg.n_streams;
		std::vector<int32_t> cnt(chains), op(chains * T.cap), cl(chains * T.cap);
		(void)hipMemcpy(cnt.data(), T.count, chains * 4, hipMemcpyDeviceToHost);
		(void)hipMemcpy(op.data(), T.open, chains * T.cap * 4, hipMemcpyDeviceToHost);
		(void)hipMemcpy(cl.data(), T.close, chains * T.cap * 4, hipMemcpyDeviceToHost);
		const int M = n_blocks * kBlockDec;
		for (int a = 0; a < c->launch.n_active; a++) {
			long hist[16] = { 0 }, nwin = 0, tot = 0;
			for (int s = 0; s < c->cfg.n_streams; s++) {
				const size_t ch = (size_t)a * c->cfg.n_streams + s;
				for (int j = 0; j < cnt[ch]; j++) {
					const int last = cl[ch * T.cap + j] < M ? cl[ch * T.cap + j] : M - 1;
					const int n = last - op[ch * T.cap + j] + 1;
					int b = 0;
					while ((256 << b) <= n && b < 15)
						b++;
					hist[b]++;
					nwin++;
					tot += n;
				}
			}
			fprintf(stderr, "WINHIST slot %d kind %d window %d: %ld windows, %ld samples (%.1f %% of the submit);", a, c->launch.params[a].kind,
				c->launch.params[a].window, nwin, tot, 100.0 * tot / ((double)M * c->cfg.n_streams));
			for (int b = 0; b < 16; b++)
				if (hist[b])
					fprintf(stderr, " <%d:%ld", 256 << b, hist[b]);
			fprintf(stderr, "\n");
		}
	}
	if (TFREC_KNOB_STR("DEBUG_CONVHIST") && c->submit_seq == 3 && !(c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS)) {
		// (debug: after how many 32-sample slots the first repair run of a biquad segment -- started from the end state of the
		// segment before -- became bit-identical to the segment's own run from a zero state: the convergence-time distribution
		// of the speculation, per chain; a build with -DTFREC_AMD_CK_EVERY=1 resolves it to one slot)
		(void)hipDeviceSynchronize();
		const WinTables &T = c->win[set];
		const size_t chains = (size_t)c->launch.n_active * c->cfg.n_streams;
		std::vector<int32_t> fx(chains * T.segcap), vt(chains);
		(void)hipMemcpy(fx.data(), T.segfix, fx.size() * 4, hipMemcpyDeviceToHost);
		(void)hipMemcpy(vt.data(), T.vtotal, chains * 4, hipMemcpyDeviceToHost);
		for (int a = 0; a < c->launch.n_active; a++) {
			if (c->launch.params[a].kind == 0)
				continue;
			std::vector<long> hist(kSegSlots + 1, 0);
			long nseg = 0, never = 0;
			for (int s = 0; s < c->cfg.n_streams; s++) {
				const size_t ch = (size_t)a * c->cfg.n_streams + s;
				const int ns = (vt[ch] + kSegSlots - 1) / kSegSlots;
				for (int k = 1; k < ns; k++) {
					if (vt[ch] - k * kSegSlots < kSegSlots)
						continue;  // (a chain's short last segment says nothing)
					const int v = fx[ch * T.segcap + k];
					nseg++;
					if (v & kSegConverged)
						hist[std::min(kSegSlots, v & ~(kSegConverged | kSegRan))]++;
					else
						never++;
				}
			}
			fprintf(stderr, "CONVHIST slot %d (window %d, segments of %d slots): %ld full segments, %ld not converged at their end; converged within n slots:",
				c->launch.slot[a], c->launch.params[a].window, kSegSlots, nseg, never);
			long cum = 0;
			for (int n = 1; n <= kSegSlots; n++) {
				cum += hist[n];
				if (n == 4 || n == 8 || n == 12 || n == 16 || n == 20 || n == 24 || n == 32 || n == 40 || n == 48 || n == 64 || n == 96 ||
				    n == 128 || n == 192 || n == 256 || n == 384 || n == 512 || n == 1024)
					fprintf(stderr, " %d:%.4f", n, nseg ? (double)cum / nseg : 0.0);
			}
			fprintf(stderr, "\n");
		}
	}
	// the drain's copies, queued now
	hipStream_t cpy = c->cq ? c->cq : c->cp;
	for (auto &e : c->done[set])
		HIPCHK(hipStreamWaitEvent(cpy, e, 0));
	c->copied_n[set] = std::min<uint32_t>(c->copy_guess, (uint32_t)c->cfg.max_events);
	{  // header, overflow flag and the first copied_n events in one go
		static_assert(kEvHeader % 16 == 0 && sizeof(tfrec_amd_event) % 16 == 0, "drain_copy_kernel moves 16 bytes per lane");
		const size_t bytes = kEvHeader + (size_t)c->copied_n[set] * sizeof(tfrec_amd_event);
		static const int copy_kernel = TFREC_KNOB_INT("COPY_KERNEL", 1, 0, 1);  // (0: hipMemcpyAsync, as until round 6)
		if (copy_kernel) {
			const size_t n16 = bytes / 16;
			static const int copy_blocks = TFREC_KNOB_INT("COPY_BLOCKS", 256, 1, 4096);
			const unsigned blocks = (unsigned)std::min<size_t>((size_t)copy_blocks, (n16 + 255) / 256);
			hipLaunchKernelGGL(tfrec::drain_copy_kernel, dim3(blocks), dim3(256), 0, cpy, (const uint4 *)c->d_evblock[set],
					   (uint4 *)c->h_evblock_dev[set], n16);
			HIPCHK(hipGetLastError());
		} else {
			HIPCHK(hipMemcpyAsync(c->h_evblock[set], c->d_evblock[set], bytes, hipMemcpyDeviceToHost, cpy));
		}
	}
	HIPCHK(hipEventRecord(c->copied[set], cpy));
	if (timing)
		c->timed = true;
	c->inflight++;
	c->last_set = set;
	c->tail_sel ^= 1;
	c->sample_base += (long long)n_blocks * kBlockDec;
	c->last_blocks = n_blocks;
	guard.ok = true;
	return TFREC_AMD_OK;
}

int tfrec_amd_submit_device(tfrec_amd_ctx *c, const void *d_iq, size_t stride, int n_blocks, void *hip_stream)
{
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = submit_common(c, d_iq, stride, n_blocks, hip_stream, false);
	if (c)
		c->hp_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return rc;
}

static int submit_host_impl(tfrec_amd_ctx *c, const uint8_t *h_iq, size_t stride, int n_blocks);

int tfrec_amd_submit_host(tfrec_amd_ctx *c, const uint8_t *h_iq, size_t stride, int n_blocks)
{
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = submit_host_impl(c, h_iq, stride, n_blocks);
	if (c)
		c->hp_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return rc;
}

static int submit_host_impl(tfrec_amd_ctx *c, const uint8_t *h_iq, size_t stride, int n_blocks)
{
	if (!c || !h_iq || n_blocks < 1 || n_blocks > c->cfg.max_blocks)
		return TFREC_AMD_E_INVAL;
	const size_t row = (size_t)n_blocks * (c->in10x ? TFREC_AMD_BLOCK_BYTES_10X : TFREC_AMD_BLOCK_BYTES);
	if (c->cfg.n_streams > 1 && stride < row)
		return TFREC_AMD_E_INVAL;
	if (c->inflight >= kSets) {
		snprintf(g_err, sizeof(g_err), "%d submits are waiting to be drained: call tfrec_amd_drain_events first", kSets);
		return TFREC_AMD_E_STATE;
	}
	HIPCHK(hipSetDevice(c->cfg.device));
	const int set = (c->head + c->inflight) % kSets;  // the set's previous user has been drained: its staging buffer is free
	const size_t need = row * (size_t)c->cfg.n_streams;
	if (c->stage_bytes[set] < need) {
		(void)hipFree(c->d_stage[set]);
		c->d_stage[set] = nullptr;
		c->stage_bytes[set] = 0;
		if (hipMalloc((void **)&c->d_stage[set], need) != hipSuccess)
			return TFREC_AMD_E_NOMEM;
		c->stage_bytes[set] = need;
	}
	// asynchronous on the front-end stream when h_iq is pinned (tfrec_amd_host_alloc); pageable memory is staged
	// by the runtime before the call returns
	HIPCHK(hipMemcpy2DAsync(c->d_stage[set], row, h_iq, stride, row, (size_t)c->cfg.n_streams, hipMemcpyHostToDevice, c->fs));
	return submit_common(c, c->d_stage[set], row, n_blocks, nullptr, true);
}

void *tfrec_amd_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
		return nullptr;
	return p;
}

void tfrec_amd_host_free(void *p)
{
	if (p)
		(void)hipHostFree(p);
}

int tfrec_amd_sync(tfrec_amd_ctx *c)
{
	if (!c)
		return TFREC_AMD_E_INVAL;
	HIPCHK(hipSetDevice(c->cfg.device));
	for (hipStream_t st : { c->fs, c->k2, c->kw, c->cs, c->cz, c->fq, c->ks, c->cq, c->aux, c->vx, c->t1, c->cp })
		if (st)
			HIPCHK(hipStreamSynchronize(st));
	return TFREC_AMD_OK;
}

int tfrec_amd_pending_events(tfrec_amd_ctx *c, int *n)
{
	if (!c || !n)
		return TFREC_AMD_E_INVAL;
	*n = 0;
	if (c->poisoned) {  // (copied[head] may never have been recorded: synchronising on it would succeed at once)
		snprintf(g_err, sizeof(g_err), "an earlier submit failed half way: the context must be recreated");
		return TFREC_AMD_E_STATE;
	}
	if (c->inflight == 0)
		return TFREC_AMD_OK;
	HIPCHK(hipSetDevice(c->cfg.device));
	HIPCHK(hipEventSynchronize(c->copied[c->head]));  // the oldest submit not yet drained
	const EventBuf eb = *c->h_eb[c->head];
	*n = (int)(std::min(eb.count, eb.capacity) - std::min(eb.dead, std::min(eb.count, eb.capacity)));  // (retracted events are not reported)
	return eb.count > eb.capacity ? TFREC_AMD_E_OVERFLOW : TFREC_AMD_OK;
}

int tfrec_amd_drain_events(tfrec_amd_ctx *c, tfrec_amd_event *out, int cap, int *n_out)
{
	if (!c || !n_out || cap < 0 || (cap > 0 && !out))
		return TFREC_AMD_E_INVAL;
	*n_out = 0;
	if (c->poisoned) {
		snprintf(g_err, sizeof(g_err), "an earlier submit failed half way: the context must be recreated");
		return TFREC_AMD_E_STATE;
	}
	if (c->inflight == 0)
		return TFREC_AMD_OK;
	HIPCHK(hipSetDevice(c->cfg.device));
	const int set = c->head;  // the oldest submit not yet drained; a younger one may still be running
	const auto hp0 = std::chrono::steady_clock::now();
	HIPCHK(hipEventSynchronize(c->copied[set]));  // the chains' ends and the copies queued by the submit
	const auto hp1 = std::chrono::steady_clock::now();
	const EventBuf eb = *c->h_eb[set];
	const uint32_t have = std::min(eb.count, eb.capacity);
	bool overflow = eb.count > eb.capacity;
	tfrec_amd_event *tmp = c->h_events[set];
	if (have > c->copied_n[set])  // more events than the submit guessed (not on cp: the copies of younger submits wait there)
		HIPCHK(hipMemcpy(tmp + c->copied_n[set], c->d_events[set] + c->copied_n[set],
				 (size_t)(have - c->copied_n[set]) * sizeof(tfrec_amd_event), hipMemcpyDeviceToHost));
	c->copy_guess = std::max<uint32_t>(c->copy_guess_min, 2 * have);
	uint32_t live = have;
	if (eb.dead) {  // a WHB stream's speculative events that the exact kernel replaced (rare): never reported
		live = 0;
		for (uint32_t i = 0; i < have; i++)
			if (tmp[i].status != kStatusDead)
				tmp[live++] = tmp[i];
	}
	c->head = (c->head + 1) % kSets;
	c->inflight--;
	c->last_drained = set;
	account_fm_log(&c->fm, eb);
	{
		int32_t wov = 0;
		memcpy(&wov, c->h_evblock[set] + kEvOverflowOff, 4);
		if (wov) {  // cannot happen (cap is the worst case); reported rather than ignored
			snprintf(g_err, sizeof(g_err), "window table overflow");
			return TFREC_AMD_E_STATE;
		}
	}
	const auto hp2 = std::chrono::steady_clock::now();
	// Order: (stream, slot, seq, BITS chunks before their flush, end_sample, offset).  The events of a stream are few:
	// bucket by stream (counting sort on indices), then order each bucket.
	auto before = [](const tfrec_amd_event &a, const tfrec_amd_event &b) {
		if (a.slot != b.slot)
			return a.slot < b.slot;
		if (a.seq != b.seq)
			return a.seq < b.seq;
		// TFREC_AMD_F_BITS: the bit chunks of a flush come before it, in the order the bits were produced
		const bool ab = a.status == TFREC_AMD_STATUS_BITS, bb = b.status == TFREC_AMD_STATUS_BITS;
		if (ab != bb)
			return ab;
		if (a.end_sample != b.end_sample)
			return a.end_sample < b.end_sample;
		return a.offset < b.offset;
	};
	const uint32_t ns = (uint32_t)c->cfg.n_streams;
	std::vector<uint32_t> &idx = c->sort_idx, &start = c->sort_start;
	idx.resize(live);
	start.assign(ns + 1, 0u);
	for (uint32_t i = 0; i < live; i++)
		start[std::min(tmp[i].stream, ns - 1) + 1]++;
	for (uint32_t s = 0; s < ns; s++)
		start[s + 1] += start[s];
	{
		std::vector<uint32_t> fill(start.begin(), start.end() - 1);
		for (uint32_t i = 0; i < live; i++)
			idx[fill[std::min(tmp[i].stream, ns - 1)]++] = i;
	}
	for (uint32_t s = 0; s < ns; s++)
		std::sort(idx.begin() + start[s], idx.begin() + start[s + 1],
			  [&](uint32_t x, uint32_t y) { return before(tmp[x], tmp[y]); });
	uint32_t ncopy = live;
	if (ncopy > (uint32_t)cap) {
		ncopy = (uint32_t)cap;
		overflow = true;
	}
	for (uint32_t i = 0; i < ncopy; i++)
		out[i] = tmp[idx[i]];
	*n_out = (int)ncopy;
	const auto hp3 = std::chrono::steady_clock::now();
	c->hp_wait += std::chrono::duration<double>(hp1 - hp0).count();
	c->hp_copy += std::chrono::duration<double>(hp2 - hp1).count();
	c->hp_sort += std::chrono::duration<double>(hp3 - hp2).count();
	c->hp_n++;
	return overflow ? TFREC_AMD_E_OVERFLOW : TFREC_AMD_OK;
}

int tfrec_amd_read_stage0(tfrec_amd_ctx *c, int stream, int16_t *out, size_t n_pairs)
{
	if (!c || !out || !c->in10x || stream < 0 || stream >= c->cfg.n_streams ||
	    n_pairs > (size_t)c->last_blocks * 4 * kBlockDec)
		return TFREC_AMD_E_INVAL;
	int rc = tfrec_amd_sync(c);
	if (rc)
		return rc;
	HIPCHK(hipMemcpy(out, c->d_in16[c->last_set] + (size_t)stream * c->in16_stride, n_pairs * sizeof(uint32_t),
			 hipMemcpyDeviceToHost));
	return TFREC_AMD_OK;
}

int tfrec_amd_read_decimated(tfrec_amd_ctx *c, int stream, int16_t *out, size_t n_pairs)
{
	if (!c || !out || stream < 0 || stream >= c->cfg.n_streams || n_pairs > (size_t)c->last_blocks * kBlockDec)
		return TFREC_AMD_E_INVAL;
	int rc = tfrec_amd_sync(c);
	if (rc)
		return rc;
	HIPCHK(hipMemcpy(out, c->d_dec[c->last_set] + (size_t)stream * c->dec_stride, n_pairs * sizeof(uint32_t),
			 hipMemcpyDeviceToHost));
	return TFREC_AMD_OK;
}

int tfrec_amd_atan_uncertain(tfrec_amd_ctx *c, uint64_t *n)
{
	if (!c || !n)
		return TFREC_AMD_E_INVAL;
	if (c->poisoned)
		return TFREC_AMD_E_STATE;
	int rc = tfrec_amd_sync(c);
	if (rc)
		return rc;
	*n = c->fm.resolved;
	for (int k = 0; k < c->inflight; k++) {  // submits not drained yet
		EventBuf eb;
		HIPCHK(hipMemcpy(&eb, c->d_eb[(c->head + k) % kSets], sizeof(eb), hipMemcpyDeviceToHost));
		*n += eb.uncertain;
	}
	return TFREC_AMD_OK;
}

int tfrec_amd_get_fm_stats(tfrec_amd_ctx *c, tfrec_amd_fm_stats *out)
{
	if (!c || !out)
		return TFREC_AMD_E_INVAL;
	if (c->poisoned)
		return TFREC_AMD_E_STATE;
	memset(out, 0, sizeof(*out));
	out->resolved = c->fm.resolved;
	out->host_verified = c->fm.verified;
	out->host_mismatch = c->fm.mismatch;
	out->undecidable = c->fm.undecidable;
	return TFREC_AMD_OK;
}

int tfrec_amd_fm_dev_probe(int device, int kind, const void *quads_v, size_t n, int32_t *out, tfrec_amd_fm_stats *stats)
{
	const int32_t *quads = (const int32_t *)quads_v;
	if (!quads || !out || n == 0 || n > (1u << 26) || kind < 0 || kind > 2)
		return TFREC_AMD_E_INVAL;
	HIPCHK(hipSetDevice(device));
	int32_t *d_q = nullptr, *d_o = nullptr;
	EventBuf *d_eb = nullptr;
	int rc = TFREC_AMD_OK;
	FmTotals tmp;
	if (hipMalloc((void **)&d_q, n * 16) != hipSuccess || hipMalloc((void **)&d_o, n * 4) != hipSuccess ||
	    hipMalloc((void **)&d_eb, sizeof(EventBuf)) != hipSuccess)
		rc = TFREC_AMD_E_NOMEM;
	// In pieces, so that the log (the first kFmLogCap slow-path decisions of a launch) does not saturate early; a caller
	// that wants EVERY sample checked compares `out` with its own reference.
	const size_t piece = 4096;
	if (rc == TFREC_AMD_OK && hipMemcpy(d_q, quads, n * 16, hipMemcpyHostToDevice) != hipSuccess)
		rc = TFREC_AMD_E_HIP;
	for (size_t o = 0; o < n && rc == TFREC_AMD_OK; o += piece) {
		const size_t m = std::min(piece, n - o);
		EventBuf eb;
		if (hipMemset(d_eb, 0, sizeof(EventBuf)) != hipSuccess || launch_fm_probe(nullptr, d_q + 4 * o, m, d_o + o, d_eb, kind) != hipSuccess ||
		    hipMemcpy(&eb, d_eb, sizeof(eb), hipMemcpyDeviceToHost) != hipSuccess)
			rc = hip_fail(hipGetLastError(), "fm_probe");
		else
			account_fm_log(&tmp, eb);
	}
	if (rc == TFREC_AMD_OK && hipMemcpy(out, d_o, n * 4, hipMemcpyDeviceToHost) != hipSuccess)
		rc = TFREC_AMD_E_HIP;
	(void)hipFree(d_q);
	(void)hipFree(d_o);
	(void)hipFree(d_eb);
	if (stats) {
		memset(stats, 0, sizeof(*stats));
		stats->resolved = tmp.resolved;
		stats->host_verified = tmp.verified;
		stats->host_mismatch = tmp.mismatch;
		stats->undecidable = tmp.undecidable;
	}
	return rc;
}

int tfrec_amd_iir_probe(int device, double cutoff, int form, const double *in, size_t n, double *out)
{
	if (!in || !out || n == 0 || n > (1u << 24) || form < 0 || form > 1 || !(cutoff > 0.0 && cutoff < 0.5))
		return TFREC_AMD_E_INVAL;
	HIPCHK(hipSetDevice(device));
	double *d_in = nullptr, *d_out = nullptr;
	int rc = TFREC_AMD_OK;
	if (hipMalloc((void **)&d_in, n * 8) != hipSuccess || hipMalloc((void **)&d_out, n * 8) != hipSuccess)
		rc = TFREC_AMD_E_NOMEM;
	if (rc == TFREC_AMD_OK && (hipMemcpy(d_in, in, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
				   launch_iir_probe(nullptr, d_in, n, biquad_coef(cutoff), d_out, form) != hipSuccess ||
				   hipMemcpy(out, d_out, n * 8, hipMemcpyDeviceToHost) != hipSuccess))
		rc = hip_fail(hipGetLastError(), "iir_probe");
	(void)hipFree(d_in);
	(void)hipFree(d_out);
	return rc;
}

int tfrec_amd_read_thresh(tfrec_amd_ctx *c, int stream, int *thresh)
{
	if (!c || !thresh || stream < 0 || stream >= c->cfg.n_streams)
		return TFREC_AMD_E_INVAL;
	if (!c->d_fsk) {
		*thresh = c->cfg.thresh;
		return TFREC_AMD_OK;
	}
	int rc = tfrec_amd_sync(c);
	if (rc)
		return rc;
	FskState f;
	HIPCHK(hipMemcpy(&f, c->d_fsk + stream, sizeof(f), hipMemcpyDeviceToHost));
	*thresh = f.thresh;
	return TFREC_AMD_OK;
}

int tfrec_amd_get_timings(tfrec_amd_ctx *c, tfrec_amd_timings *out)
{
	if (!c || !out)
		return TFREC_AMD_E_INVAL;
	if (!c->timed)
		return TFREC_AMD_E_STATE;
	// the most recently drained submit; before the first drain: the oldest one in flight
	const int set = c->last_drained >= 0 ? c->last_drained : c->head;
	hipEvent_t *ev = c->ev[set], *tev = c->tev[set];
	for (auto &e : c->done[set])
		HIPCHK(hipEventSynchronize(e));
	HIPCHK(hipEventElapsedTime(&out->frontend_ms, ev[0], ev[3]));
	if (TFREC_KNOB_STR("HOST_PROF") && c->hp_n > 2) {  // idle time of the front-end stream between two submits' front ends
		float gap = 0, total = 0;
		const int next = (set + 1) % kSets;  // (in flight: its front end started long ago)
		if (hipEventElapsedTime(&gap, ev[3], c->ev[next][0]) == hipSuccess && hipEventElapsedTime(&total, ev[0], tev[20]) == hipSuccess &&
		    gap > -1000 && gap < 1000) {
			float s2s = 0;
			if (hipEventElapsedTime(&s2s, ev[0], c->ev[next][0]) == hipSuccess)
				c->hp_s2s += s2s;
			c->hp_gap += gap;
			c->hp_lat += total;
			c->hp_gap_n++;
		}
		(void)hipGetLastError();
	}
	HIPCHK(hipEventElapsedTime(&out->fmdev_ms, ev[3], ev[1]));
	if (c->need_fmdev && c->fmdev_k2 && !(c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS))
		HIPCHK(hipEventElapsedTime(&out->fmdev_ms, tev[24], tev[25]));
	out->windows_ms = out->spec_biquad_ms = out->repair_biquad_ms = out->fix_biquad_ms = out->slicer_ms = 0;
	out->coop_slicer_ms = out->decode_ms = out->commit_ms = 0;
	out->whb_biquad_ms = out->whb_demod_ms = out->whb_decode_ms = out->whb_commit_ms = 0;
	out->tfa1_slicer_ms = out->tfa1_coop_slicer_ms = out->tfa1_decode_commit_ms = 0;
	out->whb_verify_ms = 0;
	if (c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS) {
		HIPCHK(hipEventElapsedTime(&out->chains_ms, ev[1], ev[2]));
		HIPCHK(hipEventElapsedTime(&out->total_ms, ev[0], ev[2]));
		return TFREC_AMD_OK;
	}
	// the submit ends when the last of its three chains does
	out->chains_ms = out->total_ms = 0;
	bool has[3] = { false, false, false };  // TFA_2 family, WHB, TFA_1
	for (int a = 0; a < c->launch.n_active; a++)
		has[c->launch.params[a].kind == 1 ? 0 : (c->launch.params[a].kind == 2 ? 1 : 2)] = true;
	const int last_mark[3] = { 8, 15, 20 };
	for (int k = 0; k < 3; k++)
		if (has[k]) {
			float t = 0;
			HIPCHK(hipEventElapsedTime(&t, ev[1], tev[last_mark[k]]));
			out->chains_ms = std::max(out->chains_ms, t);
			HIPCHK(hipEventElapsedTime(&t, ev[0], tev[last_mark[k]]));
			out->total_ms = std::max(out->total_ms, t);
		}
	HIPCHK(hipEventElapsedTime(&out->windows_ms, tev[0], tev[21]));
	if (has[0]) {
		float *k2_ms[3] = { &out->spec_biquad_ms, &out->repair_biquad_ms, &out->fix_biquad_ms };
		for (int k = 0; k < 3; k++)
			HIPCHK(hipEventElapsedTime(k2_ms[k], tev[1 + k], tev[2 + k]));
		HIPCHK(hipEventElapsedTime(&out->slicer_ms, tev[23], tev[5]));
		HIPCHK(hipEventElapsedTime(&out->coop_slicer_ms, tev[5], tev[6]));
		if (c->cz)  // (split off cs: its own marks)
			HIPCHK(hipEventElapsedTime(&out->coop_slicer_ms, tev[28], tev[29]));
		HIPCHK(hipEventElapsedTime(&out->decode_ms, tev[6], tev[7]));
		HIPCHK(hipEventElapsedTime(&out->commit_ms, tev[7], tev[8]));
	}
	if (has[2]) {
		HIPCHK(hipEventElapsedTime(&out->tfa1_slicer_ms, tev[16], tev[17]));
		HIPCHK(hipEventElapsedTime(&out->tfa1_coop_slicer_ms, tev[17], tev[18]));
		HIPCHK(hipEventElapsedTime(&out->tfa1_decode_commit_ms, tev[18], tev[20]));
	}
	if (has[1]) {
		HIPCHK(hipEventElapsedTime(&out->whb_biquad_ms, tev[9], tev[12]));
		HIPCHK(hipEventElapsedTime(&out->whb_demod_ms, tev[22], tev[13]));
		if (hipEventQuery(tev[27]) == hipSuccess && hipEventElapsedTime(&out->whb_verify_ms, tev[26], tev[27]) != hipSuccess)
			out->whb_verify_ms = 0;
		(void)hipGetLastError();
		// whb_decode_ms / whb_commit_ms stay 0: those stages run in the tail of whb_demod_kernel
	}
	return TFREC_AMD_OK;
}

int tfrec_amd_get_layout(tfrec_amd_ctx *c, int *n_streams)
{
	if (!c || !n_streams)
		return TFREC_AMD_E_INVAL;
	*n_streams = (c->cfg.flags & TFREC_AMD_F_SERIAL_CHAINS) ? 2 : (c->deep ? 6 : 4);
	return TFREC_AMD_OK;
}

int tfrec_amd_get_memory(tfrec_amd_ctx *c, uint64_t *device_bytes, uint64_t *pinned_host_bytes)
{
	if (!c || !device_bytes || !pinned_host_bytes)
		return TFREC_AMD_E_INVAL;
	*device_bytes = c->dev_bytes;
	for (size_t b : c->stage_bytes)  // staging of tfrec_amd_submit_host, grown on demand
		*device_bytes += b;
	*pinned_host_bytes = c->pinned_bytes;
	return TFREC_AMD_OK;
}

int tfrec_amd_get_stats(tfrec_amd_ctx *c, tfrec_amd_stats *out)
{
	if (!c || !out)
		return TFREC_AMD_E_INVAL;
	memset(out, 0, sizeof(*out));
	if (!c->win[0].stats)
		return TFREC_AMD_OK;
	int rc = tfrec_amd_sync(c);
	if (rc)
		return rc;
	constexpr int kCounters = (int)(sizeof(tfrec_amd_stats) / sizeof(uint64_t));
	static_assert(sizeof(tfrec_amd_stats) == 11 * sizeof(uint64_t) && kCounters <= 16, "the counters are the first slots of WinTables::stats");
	for (int k = 0; k < kSets; k++) {  // the table sets count separately
		tfrec_amd_stats part;
		HIPCHK(hipMemcpy(&part, c->win[k].stats, sizeof(part), hipMemcpyDeviceToHost));
		for (int i = 0; i < kCounters; i++)
			reinterpret_cast<uint64_t *>(out)[i] += reinterpret_cast<const uint64_t *>(&part)[i];
	}
	return TFREC_AMD_OK;
}

}  // extern "C"
