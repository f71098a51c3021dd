// tfrec_amd/host/main.cpp -- tfrec_gpu: the reference's file-replay CLI on the GPU path.
//
//   tfrec_gpu [-T hexmask] [-t thresh] [-W] [-q] [-D] [-B] [-d device[,device...]] [-b blocks] [-e handler | -E handler] [-m mode]
//             -L dump.iq [-L more.iq ...]
//   tfrec_gpu [-T hexmask] -X telegrams.txt
//
// Flags keep the reference's meaning (main.cpp:63-88, 107-164): -T sensor type bit mask (hex), -t trigger
// threshold (0 = auto, the default), -W wide filter, -q quiet, -D debug,
// -e handler executed for every message, -m 1 summary at exit,
// -L raw 8-bit IQ dump as written by "tfrec -S", -X hex telegrams for the byte-level test entry
// (main.cpp:24-53).  Several -L files are processed as one batch, one stream each.
// -B (not in the reference): BITS-mode replay -- every demodulated bit goes through the decoder's own store_bit
// (gpu_engine.h), so that stdout also carries what store_bit prints ("Inverted SYNC", tfa2.cpp:294-300).
// -E handler (not in the reference): the handler is started ONCE and receives the records of all
// streams on stdin, "<stream> <id> <temp> <hum> <seq> <alarm> <rssi> <flags> <ts>" per line, one write per batch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "gpu_engine.h"

static int replay_hex(int types, int dbg, const char *fn, const char *exec, bool batched)
{
	pipe_sink *psink = (batched && exec) ? new pipe_sink(exec) : NULL;
	batch_sink *sink = psink;
	std::vector<decoder *> decs;
	if (types & (1 << TFA_1)) decs.push_back(new sinked_decoder<tfa1_decoder>(TFA_1, &sink, 0));
	if (types & (1 << TFA_2)) decs.push_back(new sinked_decoder<tfa2_decoder>(TFA_2, &sink, 0));
	if (types & (1 << TFA_3)) decs.push_back(new sinked_decoder<tfa2_decoder>(TFA_3, &sink, 0));
	if (types & (1 << TX22)) decs.push_back(new sinked_decoder<tfa2_decoder>(TX22, &sink, 0));
	if (types & (1 << TFA_WHB)) decs.push_back(new sinked_decoder<whb_decoder>(TFA_WHB, &sink, 0));
	FILE *fd = fopen(fn, "r");
	if (!fd) {
		perror("Can't open message file");
		return 1;
	}
	char line[2048];
	while (fgets(line, sizeof(line), fd)) {
		if (line[0] == '#')
			continue;
		uint8_t buf[512];
		int len = 0;
		for (char *tok = strtok(line, " \t\r\n"); tok && len < (int)sizeof(buf); tok = strtok(NULL, " \t\r\n"))
			buf[len++] = (uint8_t)strtol(tok, NULL, 16);
		for (size_t k = 0; k < decs.size(); k++) {
			decs[k]->set_params(batched ? NULL : (char *)exec, 0, dbg);
			decs[k]->store_bytes(buf, len);
			decs[k]->flush(0);
			puts("");
			decs[k]->flush_storage();
		}
	}
	fclose(fd);
	delete psink;  // flushes
	return 0;
}

int main(int argc, char **argv)
{
	int types = 0x07, thresh = 0, filter = 0, dbg = 0, blocks = 16;  // defaults of main.cpp:97-105 (0 = auto)
	std::vector<int> devices;
	std::vector<std::string> dumps;
	const char *hexfile = NULL, *exec = NULL;
	bool batched = false, bits = false;
	int mode = 0;
	int c;
	while ((c = getopt(argc, argv, "T:t:WqDBd:b:L:X:e:E:m:h")) != -1) {
		switch (c) {
		case 'T': types = (int)strtol(optarg, NULL, 16); break;
		case 't': thresh = atoi(optarg); break;
		case 'W': filter = 1; break;
		case 'q': dbg = -1; break;
		case 'D': dbg++; break;
		case 'B': bits = true; break;
		case 'd':  // one ordinal or a comma-separated list: the dump files are sharded over the devices by index
			for (char *tok = strtok(optarg, ","); tok; tok = strtok(NULL, ","))
				devices.push_back(atoi(tok));
			break;
		case 'b': blocks = atoi(optarg); break;
		case 'L': dumps.push_back(optarg); break;
		case 'X': hexfile = optarg; break;
		case 'e': exec = optarg; batched = false; break;
		case 'E': exec = optarg; batched = true; break;
		case 'm': mode = atoi(optarg); break;
		default:
			fprintf(stderr, "usage: tfrec_gpu [-T hexmask] [-t thresh] [-W] [-q] [-D] [-B] [-d dev] [-b blocks] -L dump [-L dump ...] | -X hexfile\n");
			return c == 'h' ? 0 : 1;
		}
	}
	setvbuf(stdout, NULL, _IOFBF, 1 << 16);
	if (hexfile)
		return replay_hex(types, dbg, hexfile, exec, batched);
	if (dumps.empty()) {
		fprintf(stderr, "tfrec_gpu: need -L <dumpfile> or -X <hexfile>\n");
		return 1;
	}
	if (thresh < 0) {
		fprintf(stderr, "tfrec_gpu: -t must be >= 0 (0 = auto)\n");
		return 1;
	}
	gpu_engine e(dumps, types, thresh, filter, dbg, devices, blocks);
	if (exec || mode)
		e.set_handler(exec, batched, mode);
	e.set_bits_replay(bits);
	int rc = e.run();
	fflush(stdout);
	return rc ? 2 : 0;
}
