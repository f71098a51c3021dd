// tfrec_amd/host/plugin.h -- host-side mirror of the reference's plugin surface for the hot path.
//
// Same class names, virtuals, members and call conventions as baycom/tfrec's decoder.h:21-73: the adapter
// (gpu_engine.cpp, main.cpp) uses nothing else, so it compiles unchanged against the reference's OWN headers and links
// with the reference's own decoder objects (-DTFREC_AMD_REFERENCE_PLUGINS, tests/test_reference_link_cpu.py) -- this
// mirror only exists because the reference's sources do not travel to the GPU box.
// The implementations in this directory are written from the protocol documentation and the behaviour
// pinned by tests/golden (not copied); see INTEGRATION.md for the adapter contract.
#ifndef TFREC_AMD_HOST_PLUGIN_H
#define TFREC_AMD_HOST_PLUGIN_H

#include <stdint.h>
#include <time.h>

#include <map>
#include <string>

// decoder.h:11-19
enum sensor_e {
	TFA_1 = 0,  // IT+ KlimaLogg Pro, NRZS 38400 bit/s
	TFA_2,      // IT+ 17240 bit/s
	TFA_3,      // IT+ 9600 bit/s
	TX22,       // LaCrosse TX22, 8842 bit/s
	TFA_WHP,    // (unused by the reference)
	TFA_WHB,    // TFA WeatherHub, 6000 bit/s
	FIREANGEL = 0x20
};

// decoder.h:21-31
typedef struct {
	sensor_e type;
	uint64_t id;
	double temp;
	double humidity;
	int alarm;
	int flags;
	int sequence;
	time_t ts;
	int rssi;
} sensordata_t;

// decoder.h:33-59
class decoder {
public:
	decoder(sensor_e _type);
	virtual ~decoder() {}
	void set_params(char *_handler, int _mode, int _dbg);
	virtual void store_bit(int bit);
	virtual void flush(int rssi, int offset = 0);
	virtual void store_data(sensordata_t &d);
	virtual void execute_handler(sensordata_t &d);
	virtual void flush_storage(void);
	virtual int has_sync(void) { return synced; }
	int count(void) { return (int)data.size(); }
	sensor_e get_type(void) { return type; }
	virtual void store_bytes(uint8_t *d, int len);

protected:
	int dbg;
	int bad;
	int synced;
	sensor_e type;
	uint8_t rdata[256];
	int byte_cnt;

private:
	char *handler;
	int mode;
	std::map<uint64_t, sensordata_t> data;
};

// decoder.h:61-73.  On the GPU path the demodulators run in HIP; this class only keeps the (decoder*)
// association and the virtual signature for code that iterates a vector<demodulator*> like main.cpp:45-49.
class demodulator {
public:
	demodulator(decoder *_dec);
	virtual ~demodulator() {}
	virtual void start(int len);
	virtual void reset(void) {}
	virtual int demod(int thresh, int pwr, int index, int16_t *iq);
	decoder *dec;

protected:
	int last_bit_idx;
};

// crc8.h / crc32.h
class crc8 {
public:
	crc8(int poly);
	uint8_t calc(uint8_t *data, int len);

private:
	uint8_t lookup[256];
};

class crc32 {
public:
	crc32(uint32_t poly);
	uint32_t calc(uint8_t *data, int len, uint32_t init = 0);

private:
	uint32_t lookup[256];
};

// tfa1.h:10-23, tfa2.h:12-28, whb.h:12-42: the telegram parsers (flush) -- store_bit() is the bit-level sync
// search; on the GPU path bytes arrive through store_bytes() instead, but store_bit() is kept functional.
class tfa1_decoder : public decoder {
public:
	tfa1_decoder(sensor_e _type);
	void store_bit(int bit);
	void flush(int rssi, int offset = 0);

private:
	void debug_header(int nbytes, const char *tail);
	uint32_t sr;
	int sr_cnt;
	int snum;  // telegrams printed in debug mode (tfa1.h:20)
	crc8 crc;
};

class tfa2_decoder : public decoder {
public:
	tfa2_decoder(sensor_e type = TFA_2);
	void store_bit(int bit);
	void flush(int rssi, int offset = 0);

private:
	void flush_tfa(int rssi, int offset);
	void flush_tx22(int rssi, int offset);
	void rearm();
	void debug_header(int nbytes, const char *tail);
	int invert;
	uint32_t sr;
	int sr_cnt;
	int snum;
	crc8 crc;
};

class whb_decoder : public decoder {
public:
	whb_decoder(sensor_e type = TFA_WHB);
	void store_bit(int bit);
	void flush(int rssi, int offset = 0);

private:
	void emit(uint64_t id, int sub, double temp, double hum, int seq, int rssi);
	void payload(uint32_t stype, const uint8_t *msg, uint64_t id, int rssi);
	uint32_t sr;
	int sr_cnt;
	int snum;
	crc32 crc;
	uint32_t raw_hist;  // last raw bits (whb.cpp:568-580 reduces to out[t] = b[t]^b[t-12]^b[t-17])
};

#endif
