#!/bin/bash
# A/B at the DRIVER's command (20 timed steps, 5 warm-up: the FIFO's filling and draining count) and at 100 steps (steady
# state), for every library build under tfrec_amd/ab/*.so plus the default, or for environment settings:
#   profiles/ab_driver.sh <rounds>                       -> libraries
#   profiles/ab_driver.sh <rounds> "<VAR=val ..>" ...     -> environment settings with the default library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rounds=${1:-3}
shift
run() {  # $1 = label, $2 = env assignments, $3 = steps
	env $2 python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --steps $3 --warmup 5 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:5]
print('%-46s steps %3d %7.3f ms/step steady %.3f (min %.2f med %.2f max %.2f) frac %.4f  %s' % ('$1', $3, j['ms_per_step'], j['ms_per_step_steady'] or 0, j['ms_min'], j['ms_median'], j['ms_max'], j['roofline']['frac'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
}
for r in $(seq $rounds); do
	if [ $# -gt 0 ]; then
		for cfg in "$@"; do run "$cfg" "$cfg" 20; run "$cfg" "$cfg" 100; done
	else
		for lib in $R/tfrec_amd/libtfrec_amd.so $(ls $R/tfrec_amd/ab/*.so 2>/dev/null); do
			run "$(basename $lib)" "TFREC_AMD_LIB=$lib" 20; run "$(basename $lib)" "TFREC_AMD_LIB=$lib" 100
		done
	fi
done
