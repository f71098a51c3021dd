/* oracle/fm_boundary.c -- TEST INFRASTRUCTURE: search int16 quadruples (ar, aj, br, bj) whose discriminator angle
 * atan2(cj, cr) * 16384 / pi (cr + i cj = a * conj(b), dsp_stuff.cpp:284-292) lies within `tol` of an integer -- the
 * inputs on which fm_dev's truncation is hardest to get right.  For random a and every k the direction
 * arg(a) - k pi / 16384 is approximated by continued-fraction convergents b with |b| components <= bmax.
 *   fm_boundary <seed> <n_a> <bmax> <tol>   -> binary int32[4] records on stdout, summary on stderr */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static uint64_t rng;
static uint32_t rnd(void)
{
	rng ^= rng << 13;
	rng ^= rng >> 7;
	rng ^= rng << 17;
	return (uint32_t)(rng >> 16);
}

int main(int argc, char **argv)
{
	if (argc < 5)
		return 1;
	rng = strtoull(argv[1], 0, 10) * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
	const int n_a = atoi(argv[2]), bmax = atoi(argv[3]);
	const long double tol = strtold(argv[4], 0);
	const long double pi = acosl(-1.0L), scale = 16384.0L / pi;
	size_t found = 0;
	for (int ia = 0; ia < n_a; ia++) {
		const int amax = (ia & 1) ? 32767 : 8191;
		int ar = (int)(rnd() % (2 * amax + 1)) - amax, aj = (int)(rnd() % (2 * amax + 1)) - amax;
		if (!ar && !aj)
			ar = 1;
		const long double arga = atan2l((long double)aj, (long double)ar);
		for (int k = -16383; k <= 16383; k++) {
			if (!k)
				continue;
			const long double psi = arga - k * pi / 16384.0L;
			const long double cs = cosl(psi), sn = sinl(psi);
			const int swap = fabsl(sn) > fabsl(cs);
			const long double u = swap ? fabsl(sn) : fabsl(cs), w = swap ? fabsl(cs) : fabsl(sn);
			long double rho = w / u; /* in [0, 1] */
			/* convergents p/q of rho */
			long p0 = 0, q0 = 1, p1 = 1, q1 = 0;
			long double xr = rho;
			for (int it = 0; it < 40; it++) {
				const long double fl = floorl(xr);
				const long aq = (long)fl;
				const long p2 = aq * p1 + p0, q2 = aq * q1 + q0;
				if (q2 > bmax)
					break;
				p0 = p1; q0 = q1; p1 = p2; q1 = q2;
				const long double fr = xr - fl;
				if (fr < 1e-18L)
					break;
				xr = 1.0L / fr;
			}
			for (int c = 0; c < 2; c++) {
				const long p = c ? p0 : p1, q = c ? q0 : q1;
				if (q <= 0 || q > bmax || p > bmax)
					continue;
				long big = q, small = p;
				int br = (int)(swap ? small : big), bj = (int)(swap ? big : small);
				if (cs < 0)
					br = -br;
				if (sn < 0)
					bj = -bj;
				const long double cr = (long double)ar * br + (long double)aj * bj;
				const long double cj = (long double)aj * br - (long double)ar * bj;
				if (cr == 0 || cj == 0 || fabsl(cr) == fabsl(cj))
					continue;
				const long double v = atan2l(cj, cr) * scale;
				if (fabsl(v - rintl(v)) < tol) {
					int32_t rec[4] = { ar, aj, br, bj };
					fwrite(rec, sizeof(rec), 1, stdout);
					found++;
				}
			}
		}
	}
	fprintf(stderr, "fm_boundary: %zu records within %Lg\n", found, tol);
	return 0;
}
