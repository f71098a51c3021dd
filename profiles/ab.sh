#!/bin/bash
# A/B on ONE GPU box: bench.py with each library build under tfrec_amd/ab/*.so (plus the default build), alternating,
# <rounds> times.  usage: profiles/ab.sh <rounds> [bench args]   -> one line per run: lib, ms_per_step, min/median, dominant kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}
shift
cd $R
libs="$R/tfrec_amd/libtfrec_amd.so $(ls $R/tfrec_amd/ab/*.so 2>/dev/null)"
for r in $(seq $rounds); do
	for lib in $libs; do
		TFREC_AMD_LIB=$lib python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --steps 30 --warmup 5 --no-extra-configs "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:9]
print('%-28s %7.3f ms/step (min %.2f med %.2f)  %s' % ('$(basename $lib)', j['ms_per_step'], j['ms_min'], j['ms_median'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
	done
done
