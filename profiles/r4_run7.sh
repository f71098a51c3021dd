#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
	env $1 python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --steps $2 --warmup 5 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:5]
print('%-50s steps %3d %7.3f ms/step steady %.3f (min %.2f med %.2f max %.2f) %s' % ('$1', $2, j['ms_per_step'], j['ms_per_step_steady'] or 0, j['ms_min'], j['ms_median'], j['ms_max'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
}
for r in 1 2 3; do
for cfg in "TFREC_AMD_FMDEV_OWN=0" "TFREC_AMD_FMDEV_OWN=2" "TFREC_AMD_FMDEV_OWN=2 TFREC_AMD_PRIO=hlhhhnn" "TFREC_AMD_FMDEV_OWN=1" "TFREC_AMD_FMDEV_OWN=3 TFREC_AMD_PRIO=hnhhnnn"; do
	run "$cfg" 20
	run "$cfg" 100
done
done
