// Experiment and test knobs of the library.
//
// TWO builds of the same sources (tfrec_amd/_build.py):
//   libtfrec_amd.so      the product -- what bench.py, the adapter and the parity tests load.  Built WITHOUT
//                        TFREC_AMD_EXPERIMENTS: every knob below is its default as a compile-time constant, the what-if
//                        branches ("leave kernel group k out", results wrong) and the test hooks (forced WHB failures, a
//                        perturbed frozen average) fold away, and neither a getenv call nor a knob's name is in the binary
//                        (`strings libtfrec_amd.so | grep -c TFREC_AMD_SKIP` = 0; tests/test_cabi_cpu.py checks it).
//   libtfrec_amd_exp.so  -DTFREC_AMD_EXPERIMENTS: the knobs are read from the environment.  Loaded only by the tests that
//                        drive a hook (tests/: api.Receiver(..., experiments=True)) and by the A/B sessions under profiles/.
// The macros take the knob's name WITHOUT its TFREC_AMD_ prefix; in the product build the name is not expanded at all.
#pragma once

#include <stdlib.h>

#ifdef TFREC_AMD_EXPERIMENTS
namespace tfrec {
// integer from the environment, `dflt` when unset or outside [lo, hi]
static inline int knob_int(const char *name, int dflt, int lo = 0, int hi = 1 << 30)
{
	const char *v = getenv(name);
	const int x = v ? atoi(v) : dflt;
	return x >= lo && x <= hi ? x : dflt;
}
}  // namespace tfrec
#define TFREC_KNOB_INT(NAME, dflt, lo, hi) (::tfrec::knob_int("TFREC_AMD_" NAME, (dflt), (lo), (hi)))
#define TFREC_KNOB_STR(NAME) (getenv("TFREC_AMD_" NAME))
#define TFREC_KNOBS_BUILT 1
#else
#define TFREC_KNOB_INT(NAME, dflt, lo, hi) (dflt)
#define TFREC_KNOB_STR(NAME) (static_cast<const char *>(nullptr))
#define TFREC_KNOBS_BUILT 0
#endif
