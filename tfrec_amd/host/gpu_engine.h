// tfrec_amd/host/gpu_engine.h -- batched counterpart of the reference's engine (engine.h:21-45, engine.cpp:46-94).
//
// engine::run reads one dump file block by block and pushes every block through process_iq + fsk_demod::process.
// gpu_engine::run does the same for N dump files at once, on one or several GPUs: blocks are staged to the devices
// through the C ABI (include/tfrec_amd.h) and the decoder flush events that come back are replayed, per stream and
// in time order, into ordinary decoder objects through decoder::store_bytes + decoder::flush -- the reference's own
// test entry (main.cpp:45-49).
//
// The decoder classes are the reference's: built with -DTFREC_AMD_REFERENCE_PLUGINS -I<baycom/tfrec> this file includes
// the reference's own decoder.h / tfa1.h / tfa2.h / whb.h and the adapter links against the reference's own objects
// (INTEGRATION.md section 3); otherwise it uses the mirror in plugin.h (same declarations; the reference's sources do
// not travel to the GPU box).  Nothing here touches a decoder beyond its public reference interface.
#ifndef TFREC_AMD_HOST_GPU_ENGINE_H
#define TFREC_AMD_HOST_GPU_ENGINE_H

#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/tfrec_amd.h"
#ifdef TFREC_AMD_REFERENCE_PLUGINS
#include "decoder.h"
#include "tfa1.h"
#include "tfa2.h"
#include "whb.h"
#else
#include "plugin.h"
#endif

// Batched result sink.  The reference runs system("<handler> <args>") once per telegram
// (decoder.cpp:67-96): one fork+exec per record does not scale to thousands of streams.  With a sink the SAME argument
// string (id temp hum seq alarm rssi flags ts) goes to it instead, tagged with the stream; the engine flushes the sink
// once per batch.
class batch_sink {
public:
	virtual ~batch_sink() {}
	virtual void put(int stream, const char *args) = 0;
};

// the handler's argument list of decoder.cpp:67-96 (without the command), for a decoder of type dec_type
void tfrec_handler_args(const sensordata_t &d, sensor_e dec_type, char *out, size_t n);

// A protocol handler of the reference (Base = tfa1_decoder, tfa2_decoder, whb_decoder: unchanged) whose
// execute_handler() -- virtual in the reference, decoder.h:42 -- hands the record to the engine's sink when there is one.
template <class Base>
class sinked_decoder : public Base {
public:
	sinked_decoder(sensor_e t, batch_sink *const *sink_, int stream_) : Base(t), sink(sink_), stream(stream_) {}
	void execute_handler(sensordata_t &d)
	{
		if (*sink) {
			char args[384];
			tfrec_handler_args(d, this->get_type(), args, sizeof(args));
			(*sink)->put(stream, args);
		} else {
			Base::execute_handler(d);
		}
	}

private:
	batch_sink *const *sink;
	int stream;
};

// One long-lived handler process for ALL streams: records go to its stdin, one line each,
// "<stream> <id> <temp> <hum> <seq> <alarm> <rssi> <flags> <ts>" -- the reference's handler arguments
// prefixed with the stream index -- written once per batch.
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
	// types: -T bit mask; thresh: -t; filter: -W; dbg: -1 quiet, 0 normal, >=1 debug (main.cpp:97).
	// devices: HIP device ordinals; the streams (dump files) are sharded over them by index, contiguous ranges, no
	// exchange between devices; an ordinal may appear more than once (several contexts on one GPU).
	gpu_engine(const std::vector<std::string> &dumpfiles, int types, int thresh, int filter, int dbg,
		   const std::vector<int> &devices, int blocks_per_submit);
	~gpu_engine();
	// exec: per-telegram handler as the reference's -e (system() per record); batched: the same command started
	// once, records on its stdin (pipe_sink); mode: the reference's -m (1 = summary at the end)
	void set_handler(const char *exec, bool batched, int mode);
	// BITS-mode replay (SURVEY 8b "In BITS mode call dec->store_bit(b) per bit then flush"): the context reports every bit
	// the demodulators hand to decoder::store_bit (TFREC_AMD_F_BITS) and every flush; the decoders then run exactly as
	// inside the reference -- including what store_bit itself prints (tfa2.cpp:294-300 "Inverted SYNC").  Default off:
	// the byte-level replay (store_bytes + flush) moves 64 bytes per window instead of every bit.
	void set_bits_replay(bool on) { bits_replay = on; }
	// returns 0 on success, a TFREC_AMD_E_* code otherwise
	int run();
	// decoders of stream s in slot order (NULL for slots not registered)
	decoder *get_decoder(size_t s, int slot) { return decs[s][slot]; }
	long telegrams() const { return n_telegrams; }

private:
	void replay(const tfrec_amd_event &ev);
	std::vector<std::string> files;
	int types, thresh, filter, dbg, bps;
	std::vector<int> devices;
	std::vector<std::vector<decoder *> > decs;
	std::vector<long long> stream_samples;  // decimated samples each file really holds
	long n_telegrams;
	batch_sink *sink;  // (the decoders hold its address)
	pipe_sink *psink;
	int out_mode;
	bool bits_replay;
};

#endif
