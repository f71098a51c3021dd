#!/bin/bash
# session 6: what binds? the same number of samples per batch cut differently into streams x blocks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
for cfg in "1024 48" "4096 12" "2048 24" "512 96" "256 48" "512 48" "2048 48"; do
	set -- $cfg
	python bench.py --streams $1 --blocks $2 --steps 40 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:10]
print('%5d streams x %3d blocks: %7.3f ms/step steady %s  value %9.0f MSamples/s  parity %s/%s  %s' % ($1, $2, j['ms_per_step'], j.get('ms_per_step_steady'), j['value'], j['config']['parity_ok'], j['config']['parity_after_timed'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
" >> gpurun_out/s6/shapes.txt 2>&1
done
