#!/bin/bash
# the driver's command (20 timed steps, 5 warm-up) for each library build, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
libs="$R/tfrec_amd/libtfrec_amd.so $(ls $R/tfrec_amd/ab/*.so 2>/dev/null)"
for r in 1 2 3 4; do
	for lib in $libs; do
		TFREC_AMD_LIB=$lib python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --steps 20 --warmup 5 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:4]
print('%-22s %7.3f ms/step (min %.2f med %.2f max %.2f) frac %.4f after %s  %s' % ('$(basename $lib)', j['ms_per_step'], j['ms_min'], j['ms_median'], j['ms_max'], j['roofline']['frac'], j['config']['parity_after_timed'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
	done
done
