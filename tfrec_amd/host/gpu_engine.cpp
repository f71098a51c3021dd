// tfrec_amd/host/gpu_engine.cpp -- see gpu_engine.h.
#include "gpu_engine.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

gpu_engine::gpu_engine(const std::vector<std::string> &dumpfiles, int _types, int _thresh, int _filter, int _dbg,
		       int _device, int blocks_per_submit)
	: files(dumpfiles), types(_types), thresh(_thresh), filter(_filter), dbg(_dbg), device(_device),
	  bps(blocks_per_submit), n_telegrams(0), sink(NULL), out_mode(0)
{
	// one set of protocol handlers per stream, registered like main.cpp:173-218
	for (size_t s = 0; s < files.size(); s++) {
		std::vector<decoder *> d(TFREC_AMD_NSLOTS, (decoder *)NULL);
		if (types & (1 << TFA_1)) d[TFREC_AMD_SLOT_TFA1] = new tfa1_decoder(TFA_1);
		if (types & (1 << TFA_2)) d[TFREC_AMD_SLOT_TFA2] = new tfa2_decoder(TFA_2);
		if (types & (1 << TFA_3)) d[TFREC_AMD_SLOT_TFA3] = new tfa2_decoder(TFA_3);
		if (types & (1 << TX22)) d[TFREC_AMD_SLOT_TX22] = new tfa2_decoder(TX22);
		if (types & (1 << TFA_WHB)) d[TFREC_AMD_SLOT_WHB] = new whb_decoder(TFA_WHB);
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
		sink = new pipe_sink(exec);
	for (size_t s = 0; s < decs.size(); s++)
		for (size_t k = 0; k < decs[s].size(); k++)
			if (decs[s][k]) {
				decs[s][k]->set_params(batched ? NULL : (char *)exec, mode, dbg);
				if (sink)
					decs[s][k]->set_sink(sink, (int)s);
			}
}

gpu_engine::~gpu_engine()
{
	delete sink;
	for (size_t s = 0; s < decs.size(); s++)
		for (size_t k = 0; k < decs[s].size(); k++)
			delete decs[s][k];
}

// The adapter contract: bring the decoder's rdata[0..64) to the state the GPU decoder had,
// set byte_cnt, then let the unchanged handler do CRC, parsing, printing, store_data.
void gpu_engine::replay(const tfrec_amd_event &ev)
{
	decoder *dec = decs[ev.stream][ev.slot];
	if (!dec)
		return;
	uint8_t buf[256];
	memset(buf, 0, sizeof(buf));
	memcpy(buf, ev.rdata, 64);
	dec->store_bytes(buf, 64);
	int len = ev.byte_cnt > 256 ? 256 : ev.byte_cnt;
	dec->store_bytes(buf, len);
	int before = dec->count();
	dec->flush(tfrec_amd_rssi_db(ev.slot, ev.rssi_raw), ev.offset);
	if (ev.status == 1)
		n_telegrams++;
	(void)before;
}

// engine::run (engine.cpp:63-93) for N files at once, as a three-stage pipeline over batches of bps blocks:
//   reader thread : fread batch k+2 of every file into a pinned host buffer (three buffers in rotation)
//   GPU           : H2D copy + hot path of batch k+1 (tfrec_amd_submit_host is asynchronous on pinned memory)
//   this thread   : drain batch k's flush events and replay them into the decoders
// The C ABI's submit/drain FIFO (depth TFREC_AMD_FIFO_DEPTH = 3) is what lets batch k+1 be queued before batch k is
// drained; this loop keeps two in flight (the host side, not the GPU, bounds a file replay: DESIGN.md section 6).
int gpu_engine::run()
{
	const size_t n = files.size();
	std::vector<FILE *> fd(n, (FILE *)NULL);
	size_t max_blocks = 0;
	stream_samples.assign(n, 0);
	for (size_t s = 0; s < n; s++) {
		fd[s] = fopen(files[s].c_str(), "rb");
		if (!fd[s]) {
			perror(files[s].c_str());
			return TFREC_AMD_E_INVAL;
		}
		fseek(fd[s], 0, SEEK_END);
		const size_t blocks = (size_t)ftell(fd[s]) / TFREC_AMD_BLOCK_BYTES;  // trailing partial block dropped, engine.cpp:72-76
		fseek(fd[s], 0, SEEK_SET);
		stream_samples[s] = (long long)blocks * TFREC_AMD_BLOCK_DEC;
		max_blocks = std::max(max_blocks, blocks);
	}
	tfrec_amd_config cfg;
	memset(&cfg, 0, sizeof(cfg));
	cfg.n_streams = (int32_t)n;
	cfg.types_mask = types;
	cfg.thresh = thresh;
	cfg.filter_type = filter;
	cfg.device = device;
	cfg.max_blocks = bps;
	cfg.max_events = (int32_t)std::max<size_t>(4096, n * (size_t)bps * 64);
	cfg.flags = 0;
	tfrec_amd_ctx *ctx = NULL;
	int rc = tfrec_amd_create(&cfg, &ctx);
	if (rc) {
		fprintf(stderr, "tfrec_amd_create: %s (%s)\n", tfrec_amd_strerror(rc), tfrec_amd_last_error());
		return rc;
	}
	const size_t row = (size_t)bps * TFREC_AMD_BLOCK_BYTES;
	const size_t n_batches = (max_blocks + bps - 1) / bps;
	constexpr int kBufs = 3;
	uint8_t *host[kBufs];
	bool pinned = true;
	for (int b = 0; b < kBufs; b++) {
		host[b] = (uint8_t *)tfrec_amd_host_alloc(n * row);
		if (!host[b]) {  // no page-locked memory: the copies become synchronous, results are the same
			pinned = false;
			host[b] = (uint8_t *)malloc(n * row);
		}
	}
	// ---- reader thread: batch k goes to host[k % kBufs]; it may run at most kBufs batches ahead of the drain
	std::mutex mu;
	std::condition_variable cv;
	size_t filled = 0, drained = 0;  // batches read / batches whose buffer is free again
	std::thread reader([&]() {
		for (size_t k = 0; k < n_batches; k++) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return k < drained + kBufs; });
			}
			const int nb = (int)std::min<size_t>(bps, max_blocks - k * bps);
			uint8_t *buf = host[k % kBufs];
			for (size_t s = 0; s < n; s++) {
				uint8_t *dst = buf + s * row;
				const size_t want = (size_t)nb * TFREC_AMD_BLOCK_BYTES;
				size_t got = fread(dst, 1, want, fd[s]);
				got -= got % TFREC_AMD_BLOCK_BYTES;
				memset(dst + got, 0x80, want - got);  // a shorter file is padded with silence (its events are cut below)
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				filled = k + 1;
			}
			cv.notify_all();
		}
	});
	std::vector<tfrec_amd_event> ev(cfg.max_events);
	auto submit = [&](size_t k) -> int {
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&]() { return filled > k; });
		}
		const int nb = (int)std::min<size_t>(bps, max_blocks - k * bps);
		return tfrec_amd_submit_host(ctx, host[k % kBufs], row, nb);
	};
	if (n_batches > 0)
		rc = submit(0);
	for (size_t k = 0; k < n_batches && rc == 0; k++) {
		if (k + 1 < n_batches && (rc = submit(k + 1)) != 0)
			break;
		int nev = 0;
		rc = tfrec_amd_drain_events(ctx, ev.data(), (int)ev.size(), &nev);
		if (rc)
			break;
		{
			std::lock_guard<std::mutex> lk(mu);
			drained = k + 1;  // batch k's host buffer may be refilled
		}
		cv.notify_all();
		// per stream in time order, slots in registration order like the reference's dispatch loop (fm_demod.cpp:48-49)
		std::sort(ev.begin(), ev.begin() + nev, [](const tfrec_amd_event &a, const tfrec_amd_event &b) {
			if (a.stream != b.stream) return a.stream < b.stream;
			if (a.end_sample != b.end_sample) return a.end_sample < b.end_sample;
			return a.slot < b.slot;
		});
		for (int q = 0; q < nev; q++)
			if (ev[q].end_sample < stream_samples[ev[q].stream])
				replay(ev[q]);
		if (sink)
			sink->flush();  // the records of the whole batch in one write
	}
	if (out_mode)  // -m 1: summary at the end (decoder.cpp:98-109)
		for (size_t s = 0; s < decs.size(); s++)
			for (size_t k2 = 0; k2 < decs[s].size(); k2++)
				if (decs[s][k2])
					decs[s][k2]->flush_storage();
	if (sink)
		sink->flush();
	if (rc)
		fprintf(stderr, "tfrec_amd: %s (%s)\n", tfrec_amd_strerror(rc), tfrec_amd_last_error());
	{
		std::lock_guard<std::mutex> lk(mu);
		drained = n_batches + kBufs;  // let the reader run out after an error
	}
	cv.notify_all();
	reader.join();
	tfrec_amd_destroy(ctx);
	for (int b = 0; b < kBufs; b++) {
		if (pinned)
			tfrec_amd_host_free(host[b]);
		else
			free(host[b]);
	}
	for (size_t s = 0; s < n; s++)
		fclose(fd[s]);
	return rc;
}
