// tfrec_amd/host/gpu_engine.h -- batched counterpart of the reference's engine (engine.h:21-45, engine.cpp:46-94).
//
// engine::run reads one dump file block by block and pushes every block through process_iq + fsk_demod::process.
// gpu_engine::run does the same for N dump files at once: blocks are staged to the GPU through the C ABI
// (include/tfrec_amd.h) and the decoder flush events that come back are replayed, per stream and in time
// order, into ordinary decoder objects (plugin.h) through decoder::store_bytes + decoder::flush -- the
// reference's own test entry (main.cpp:45-49).
#ifndef TFREC_AMD_HOST_GPU_ENGINE_H
#define TFREC_AMD_HOST_GPU_ENGINE_H

#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/tfrec_amd.h"
#include "plugin.h"

// One long-lived handler process for ALL streams: records go to its stdin, one line each,
// "<stream> <id> <temp> <hum> <seq> <alarm> <rssi> <flags> <ts>" -- the reference's handler arguments
// (decoder.cpp:67-96) prefixed with the stream index -- written once per batch.
class pipe_sink : public batch_sink {
public:
	explicit pipe_sink(const char *command);
	~pipe_sink();
	void put(int stream, const char *args);
	void flush();
	long records() const { return n_records; }

private:
	FILE *pipe;
	std::string pending;
	long n_records;
};

class gpu_engine {
public:
	// types: -T bit mask; thresh: -t; filter: -W; dbg: -1 quiet, 0 normal, >=1 debug (main.cpp:97)
	gpu_engine(const std::vector<std::string> &dumpfiles, int types, int thresh, int filter, int dbg, int device,
		   int blocks_per_submit);
	~gpu_engine();
	// exec: per-telegram handler as the reference's -e (system() per record); batched: the same command started
	// once, records on its stdin (pipe_sink); mode: the reference's -m (1 = summary at the end)
	void set_handler(const char *exec, bool batched, int mode);
	// returns 0 on success, a TFREC_AMD_E_* code otherwise
	int run();
	// decoders of stream s in slot order (NULL for slots not registered)
	decoder *get_decoder(size_t s, int slot) { return decs[s][slot]; }
	long telegrams() const { return n_telegrams; }

private:
	void replay(const tfrec_amd_event &ev);
	std::vector<std::string> files;
	int types, thresh, filter, dbg, device, bps;
	std::vector<std::vector<decoder *> > decs;
	std::vector<long long> stream_samples;  // decimated samples each file really holds
	long n_telegrams;
	pipe_sink *sink;
	int out_mode;
};

#endif
