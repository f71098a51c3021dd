#!/bin/bash
# A/B of environment settings on ONE GPU box: profiles/ab_env.sh <rounds> "<VAR=val ...>" "<VAR=val ...>" ...  (bench.py, one line per run)
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-2}
shift
cd $R
for r in $(seq $rounds); do
	for cfg in "$@"; do
		env $cfg python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --steps 30 --warmup 5 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:5]
print('%-44s %7.3f ms/step (min %.2f med %.2f) parity %s  %s' % ('$cfg', j['ms_per_step'], j['ms_min'], j['ms_median'], j['config']['parity_ok'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
	done
done
