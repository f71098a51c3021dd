// tfrec_amd/host/gpu_engine.cpp -- see gpu_engine.h.
#include "gpu_engine.h"

#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

// decoder.cpp:67-96: "<id> <temp> <hum> <seq> <alarm> <rssi> <flags> <ts>"; every type but WHB folds the sensor type into
// the id, WHB prints its 52-bit id (decoder.cpp:72-91)
void tfrec_handler_args(const sensordata_t &d, sensor_e dec_type, char *out, size_t n)
{
	if (dec_type != TFA_WHB)
		snprintf(out, n, "%04" PRIx64 " %+.1f %g %i %i %i %i %li", (uint64_t)(d.id | ((uint64_t)d.type << 24)), d.temp, d.humidity,
			 d.sequence, d.alarm, d.rssi, d.flags, (long)d.ts);
	else
		snprintf(out, n, "%013" PRIx64 " %+.1f %g %i %i %i %i %li", (uint64_t)d.id, d.temp, d.humidity, d.sequence, d.alarm,
			 d.rssi, d.flags, (long)d.ts);
}

gpu_engine::gpu_engine(const std::vector<std::string> &dumpfiles, int _types, int _thresh, int _filter, int _dbg,
		       const std::vector<int> &_devices, int blocks_per_submit)
	: files(dumpfiles), types(_types), thresh(_thresh), filter(_filter), dbg(_dbg), bps(blocks_per_submit),
	  devices(_devices), n_telegrams(0), sink(NULL), psink(NULL), out_mode(0), bits_replay(false)
{
	if (devices.empty())
		devices.push_back(0);
	// one set of protocol handlers per stream, registered like main.cpp:173-218
	for (size_t s = 0; s < files.size(); s++) {
		std::vector<decoder *> d(TFREC_AMD_NSLOTS, (decoder *)NULL);
		if (types & (1 << TFA_1)) d[TFREC_AMD_SLOT_TFA1] = new sinked_decoder<tfa1_decoder>(TFA_1, &sink, (int)s);
		if (types & (1 << TFA_2)) d[TFREC_AMD_SLOT_TFA2] = new sinked_decoder<tfa2_decoder>(TFA_2, &sink, (int)s);
		if (types & (1 << TFA_3)) d[TFREC_AMD_SLOT_TFA3] = new sinked_decoder<tfa2_decoder>(TFA_3, &sink, (int)s);
		if (types & (1 << TX22)) d[TFREC_AMD_SLOT_TX22] = new sinked_decoder<tfa2_decoder>(TX22, &sink, (int)s);
		if (types & (1 << TFA_WHB)) d[TFREC_AMD_SLOT_WHB] = new sinked_decoder<whb_decoder>(TFA_WHB, &sink, (int)s);
		for (size_t k = 0; k < d.size(); k++)
			if (d[k])
				d[k]->set_params(NULL, 0, dbg);
		decs.push_back(d);
	}
}

pipe_sink::pipe_sink(const char *command) : pipe(popen(command, "w")), n_records(0)
{
	if (!pipe)
		perror(command);
}

pipe_sink::~pipe_sink()
{
	flush();
	if (pipe)
		pclose(pipe);
}

void pipe_sink::put(int stream, const char *args)
{
	char head[32];
	snprintf(head, sizeof(head), "%d ", stream);
	pending += head;
	pending += args;
	pending += '\n';
	n_records++;
}

void pipe_sink::flush()
{
	if (pipe && !pending.empty()) {
		fwrite(pending.data(), 1, pending.size(), pipe);
		fflush(pipe);
	}
	pending.clear();
}

void gpu_engine::set_handler(const char *exec, bool batched, int mode)
{
	out_mode = mode;
	if (batched && exec && *exec)
		sink = psink = new pipe_sink(exec);
	for (size_t s = 0; s < decs.size(); s++)
		for (size_t k = 0; k < decs[s].size(); k++)
			if (decs[s][k])
				decs[s][k]->set_params(batched ? NULL : (char *)exec, mode, dbg);
}

gpu_engine::~gpu_engine()
{
	delete psink;
	// (the reference's decoder has no virtual destructor -- main.cpp never frees its plugins either)
}

// The adapter contract: bring the decoder's rdata[0..64) to the state the GPU decoder had,
// set byte_cnt, then let the unchanged handler do CRC, parsing, printing, store_data.
// BITS mode: the decoder receives every bit through its own store_bit (decoder.h:39) -- it then holds rdata / byte_cnt
// by itself, and whatever store_bit prints appears as in the reference -- and every flush, without store_bytes.
void gpu_engine::replay(const tfrec_amd_event &ev)
{
	decoder *dec = decs[ev.stream][ev.slot];
	if (!dec)
		return;
	if (ev.status == TFREC_AMD_STATUS_BITS) {
		for (int k = 0; k < (int)ev.byte_cnt && k < 512; k++)
			dec->store_bit((ev.rdata[k >> 3] >> (k & 7)) & 1);
		return;
	}
	if (!bits_replay) {
		uint8_t buf[256];
		memset(buf, 0, sizeof(buf));
		memcpy(buf, ev.rdata, 64);
		dec->store_bytes(buf, 64);
		int len = ev.byte_cnt > 256 ? 256 : ev.byte_cnt;
		dec->store_bytes(buf, len);
	}
	dec->flush(tfrec_amd_rssi_db(ev.slot, ev.rssi_raw), ev.offset);
	if (ev.status == 1)
		n_telegrams++;
}

namespace {

// engine::run (engine.cpp:63-93) for the dump files [s0, s1) on ONE device, as a three-stage pipeline over batches of
// bps blocks:
//   reader thread : fread batch k+2 of every file into a pinned host buffer (three buffers in rotation)
//   GPU           : H2D copy + hot path of batch k+1 (tfrec_amd_submit_host is asynchronous on pinned memory)
//   worker thread : drain batch k's flush events and queue them for the engine's thread
// The C ABI's submit/drain FIFO (depth TFREC_AMD_FIFO_DEPTH = 4) is what lets batches k+1 .. k+3 be queued before batch
// k is drained; this loop keeps the FIFO full (one pinned host buffer per submit in flight + one being read).
struct device_worker {
	const std::vector<std::string> *files;
	size_t s0, s1;
	int device, types, thresh, filter, bps;
	uint32_t flags;     // TFREC_AMD_F_* of the context
	size_t max_blocks;  // of ALL files: every device runs the same number of batches
	int rc;
	std::atomic<bool> *abort;  // set by the engine when any worker failed: stop instead of running the whole job
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::vector<tfrec_amd_event> > out;  // batches drained, oldest first
	bool done;
	std::thread th;

	device_worker() : files(NULL), s0(0), s1(0), device(0), types(0), thresh(0), filter(0), bps(1), flags(0), max_blocks(0), rc(0), abort(NULL), done(false) {}

	void push(std::vector<tfrec_amd_event> &&ev)
	{
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&]() { return out.size() < 2; });  // the engine's thread is at most two batches behind
			out.push_back(std::move(ev));
		}
		cv.notify_all();
	}
	// next batch's events (false: the worker ended -- rc says why)
	bool pop(std::vector<tfrec_amd_event> &ev)
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&]() { return !out.empty() || done; });
		if (out.empty())
			return false;
		ev = std::move(out.front());
		out.pop_front();
		lk.unlock();
		cv.notify_all();
		return true;
	}

	void run()
	{
		rc = work();
		{
			std::lock_guard<std::mutex> lk(mu);
			done = true;
		}
		cv.notify_all();
	}

	int work()
	{
		const size_t n = s1 - s0;
		std::vector<FILE *> fd(n, (FILE *)NULL);
		for (size_t s = 0; s < n; s++) {
			fd[s] = fopen((*files)[s0 + s].c_str(), "rb");
			if (!fd[s]) {
				perror((*files)[s0 + s].c_str());
				for (size_t q = 0; q < s; q++)
					fclose(fd[q]);
				return TFREC_AMD_E_INVAL;
			}
		}
		tfrec_amd_config cfg;
		memset(&cfg, 0, sizeof(cfg));
		cfg.n_streams = (int32_t)n;
		cfg.types_mask = types;
		cfg.thresh = thresh;
		cfg.filter_type = filter;
		cfg.device = device;
		cfg.max_blocks = bps;
		// (BITS mode: every flush + a chunk per 512 bits, a slicer emits < 0.5 bit per decimated sample)
		cfg.max_events = (int32_t)std::max<size_t>(4096, n * (size_t)bps * ((flags & TFREC_AMD_F_BITS) ? 256 : 64));
		cfg.flags = flags;
		tfrec_amd_ctx *ctx = NULL;
		int r = tfrec_amd_create(&cfg, &ctx);
		if (r) {
			fprintf(stderr, "tfrec_amd_create (device %d): %s (%s)\n", device, tfrec_amd_strerror(r), tfrec_amd_last_error());
			for (size_t s = 0; s < n; s++)
				fclose(fd[s]);
			return r;
		}
		const size_t row = (size_t)bps * TFREC_AMD_BLOCK_BYTES;
		const size_t n_batches = (max_blocks + bps - 1) / bps;
		const int depth = std::max(1, std::min(tfrec_amd_fifo_depth(), TFREC_AMD_FIFO_DEPTH));
		constexpr int kBufs = TFREC_AMD_FIFO_DEPTH + 1;
		uint8_t *host[kBufs];
		bool pinned[kBufs];  // per buffer: each is released by the allocator it came from
		for (int b = 0; b < kBufs; b++) {
			host[b] = (uint8_t *)tfrec_amd_host_alloc(n * row);
			pinned[b] = host[b] != NULL;
			if (!host[b])  // no page-locked memory: this buffer's copies become synchronous, results are the same
				host[b] = (uint8_t *)malloc(n * row);
		}
		// ---- reader thread: batch k goes to host[k % kBufs]; it may run at most kBufs batches ahead of the drain
		std::mutex rmu;
		std::condition_variable rcv;
		size_t filled = 0, drained = 0;  // batches read / batches whose buffer is free again
		std::thread reader([&]() {
			for (size_t k = 0; k < n_batches; k++) {
				{
					std::unique_lock<std::mutex> lk(rmu);
					rcv.wait(lk, [&]() { return k < drained + kBufs; });
				}
				const int nb = (int)std::min<size_t>(bps, max_blocks - k * bps);
				uint8_t *buf = host[k % kBufs];
				for (size_t s = 0; s < n; s++) {
					uint8_t *dst = buf + s * row;
					const size_t want = (size_t)nb * TFREC_AMD_BLOCK_BYTES;
					size_t got = fread(dst, 1, want, fd[s]);
					got -= got % TFREC_AMD_BLOCK_BYTES;
					memset(dst + got, 0x80, want - got);  // a shorter file is padded with silence (its events are cut by the engine)
				}
				{
					std::lock_guard<std::mutex> lk(rmu);
					filled = k + 1;
				}
				rcv.notify_all();
			}
		});
		auto submit = [&](size_t k) -> int {
			{
				std::unique_lock<std::mutex> lk(rmu);
				rcv.wait(lk, [&]() { return filled > k; });
			}
			const int nb = (int)std::min<size_t>(bps, max_blocks - k * bps);
			return tfrec_amd_submit_host(ctx, host[k % kBufs], row, nb);
		};
		size_t queued = 0;
		for (size_t k = 0; k < n_batches && r == 0; k++) {
			while (queued < n_batches && queued < k + (size_t)depth && r == 0)
				r = submit(queued++);
			if (r)
				break;
			if (abort && abort->load()) {
				r = TFREC_AMD_E_STATE;
				break;
			}
			std::vector<tfrec_amd_event> ev(cfg.max_events);
			int nev = 0;
			r = tfrec_amd_drain_events(ctx, ev.data(), (int)ev.size(), &nev);
			if (r == TFREC_AMD_E_OVERFLOW) {  // the events that fit were returned; the replay goes on (those beyond are lost)
				fprintf(stderr, "tfrec_amd: device %d batch %zu: event buffer overflow, %d events kept\n", device, k, nev);
				r = 0;
			}
			if (r)
				break;
			{
				std::lock_guard<std::mutex> lk(rmu);
				drained = k + 1;  // batch k's host buffer may be refilled
			}
			rcv.notify_all();
			ev.resize(nev);
			for (int q = 0; q < nev; q++)
				ev[q].stream += (uint32_t)s0;  // index within the whole job
			push(std::move(ev));
		}
		if (r)
			fprintf(stderr, "tfrec_amd (device %d): %s (%s)\n", device, tfrec_amd_strerror(r), tfrec_amd_last_error());
		{
			std::lock_guard<std::mutex> lk(rmu);
			drained = n_batches + kBufs;  // let the reader run out after an error
		}
		rcv.notify_all();
		reader.join();
		tfrec_amd_destroy(ctx);
		for (int b = 0; b < kBufs; b++) {
			if (pinned[b])
				tfrec_amd_host_free(host[b]);
			else
				free(host[b]);
		}
		for (size_t s = 0; s < n; s++)
			fclose(fd[s]);
		return r;
	}
};

}  // namespace

// One worker (host thread + context + HIP streams) per device entry, streams sharded by index over them; this thread
// takes the devices' events batch by batch, in device = stream order, and replays them into the decoders: stdout and
// the handler records come out in the order of a single-device run whatever the number of devices.
int gpu_engine::run()
{
	const size_t n = files.size();
	size_t max_blocks = 0;
	stream_samples.assign(n, 0);
	for (size_t s = 0; s < n; s++) {
		FILE *f = fopen(files[s].c_str(), "rb");
		if (!f) {
			perror(files[s].c_str());
			return TFREC_AMD_E_INVAL;
		}
		fseek(f, 0, SEEK_END);
		const size_t blocks = (size_t)ftell(f) / TFREC_AMD_BLOCK_BYTES;  // trailing partial block dropped, engine.cpp:72-76
		fclose(f);
		stream_samples[s] = (long long)blocks * TFREC_AMD_BLOCK_DEC;
		max_blocks = std::max(max_blocks, blocks);
	}
	const size_t n_batches = (max_blocks + bps - 1) / bps;
	const size_t nd = std::min(devices.size(), n);  // never more workers than streams
	std::vector<device_worker> workers(nd);
	for (size_t d = 0; d < nd; d++) {
		device_worker &w = workers[d];
		const size_t base = n / nd, rem = n % nd;  // contiguous ranges, as evenly as possible (tfrec_amd/shard.py)
		w.s0 = d * base + std::min(d, rem);
		w.s1 = w.s0 + base + (d < rem ? 1 : 0);
		w.files = &files;
		w.device = devices[d];
		w.types = types;
		w.thresh = thresh;
		w.filter = filter;
		w.bps = bps;
		w.flags = bits_replay ? (TFREC_AMD_F_BITS | TFREC_AMD_F_ALL_FLUSHES) : 0u;
		w.max_blocks = max_blocks;
	}
	std::atomic<bool> abort(false);
	for (size_t d = 0; d < nd; d++) {
		workers[d].abort = &abort;
		workers[d].th = std::thread([&workers, d]() { workers[d].run(); });
	}
	int rc = 0;
	std::vector<tfrec_amd_event> ev;
	for (size_t k = 0; k < n_batches && rc == 0; k++) {
		for (size_t d = 0; d < nd && rc == 0; d++) {
			if (!workers[d].pop(ev)) {
				rc = workers[d].rc ? workers[d].rc : TFREC_AMD_E_STATE;
				break;
			}
			// per stream in time order, slots in registration order like the reference's dispatch loop (fm_demod.cpp:48-49).
			// BITS chunks carry the first sample of their trigger window (a chunk has no per-bit time): a window's bits
			// are replayed when it opens, its flush when it closes -- what store_bit prints keeps its place among the
			// flushes of every window that does not overlap this one.
			std::stable_sort(ev.begin(), ev.end(), [](const tfrec_amd_event &a, const tfrec_amd_event &b) {
				if (a.stream != b.stream) return a.stream < b.stream;
				if (a.end_sample != b.end_sample) return a.end_sample < b.end_sample;
				const bool ab = a.status == TFREC_AMD_STATUS_BITS, bb = b.status == TFREC_AMD_STATUS_BITS;
				if (ab != bb) return ab;
				if (a.slot != b.slot) return a.slot < b.slot;
				return ab && a.offset < b.offset;
			});
			for (size_t q = 0; q < ev.size(); q++)
				if (ev[q].end_sample < stream_samples[ev[q].stream])
					replay(ev[q]);
		}
		if (psink)
			psink->flush();  // the records of the whole batch in one write
	}
	// (after an error: tell the healthy workers to stop, and empty the queues so that they can finish)
	if (rc)
		abort.store(true);
	for (size_t d = 0; d < nd; d++) {
		while (workers[d].pop(ev)) {
		}
		workers[d].th.join();
		if (!rc)
			rc = workers[d].rc;
	}
	if (out_mode)  // -m 1: summary at the end (decoder.cpp:98-109)
		for (size_t s = 0; s < decs.size(); s++)
			for (size_t k2 = 0; k2 < decs[s].size(); k2++)
				if (decs[s][k2])
					decs[s][k2]->flush_storage();
	if (psink)
		psink->flush();
	return rc;
}
