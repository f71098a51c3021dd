// oracle/fm_resolve_check.cpp -- TEST INFRASTRUCTURE: drives the product's exact fm_dev slow path
// (tfrec_amd/csrc/fm_resolve.h, compiled here for the host) against this host's libm, the arithmetic the reference
// binary uses (dsp_stuff.cpp:284-292).
//   fm_resolve_check quads   < int32[4] records (ar, aj, br, bj)   -> per record: int32 resolved, int32 libm, double margin
//   fm_resolve_check cross   < int64[2] records (cr, cj)           -> same
// Summary on stderr.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../tfrec_amd/csrc/fm_resolve.h"

static const double kCoarse[129][4] = { TFREC_FM_COARSE_TABLE };
static const double kFine[128][4] = { TFREC_FM_FINE_TABLE };

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;
	const bool cross = !strcmp(argv[1], "cross");
	size_t n = 0, slow = 0, mism = 0, fuzz = 0;
	double min_margin = 1e300;
	for (;;) {
		double cr, cj;
		if (cross) {
			int64_t c[2];
			if (fread(c, sizeof(c), 1, stdin) != 1)
				break;
			cr = (double)c[0];
			cj = (double)c[1];
		} else {
			int32_t q[4];
			if (fread(q, sizeof(q), 1, stdin) != 1)
				break;
			cr = ((double)q[0]) * q[2] + ((double)q[1]) * q[3];
			cj = ((double)q[1]) * q[2] - ((double)q[0]) * q[3];
		}
		n++;
		const double v = atan2(cj, cr) * tfrec::kFmScale;
		const int32_t want = (int32_t)v;
		int32_t got = want;
		double margin = -1.0;
		if (cr != 0.0 && cj != 0.0 && fabs(cr) != fabs(cj) && fabs(v - rint(v)) < 1e-9) {
			slow++;
			got = tfrec::fm_dev_resolve(cr, cj, v, kCoarse, kFine, &margin);
			if (margin < min_margin)
				min_margin = margin;
			if (margin < 0.06)
				fuzz++;
			if (got != want) {
				mism++;
				fprintf(stderr, "MISMATCH cr %.0f cj %.0f v %.17g got %d want %d margin %.4g ulp\n", cr, cj, v, got, want,
					margin);
			}
		}
		fwrite(&got, 4, 1, stdout);
		fwrite(&want, 4, 1, stdout);
		fwrite(&margin, 8, 1, stdout);
	}
	fprintf(stderr, "{\"records\": %zu, \"slow_path\": %zu, \"mismatch\": %zu, \"fuzz_band\": %zu, \"min_margin_ulps\": %.4g}\n", n, slow,
		mism, fuzz, min_margin);
	return mism ? 2 : 0;
}
