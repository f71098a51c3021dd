#!/bin/bash
# session 42: the WHB chain by itself (-T 20), WHB + TFA_1 (-T 21), WHB + the TFA_2 family (-T 2e), all five (-T 2f): where the period comes from
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s42
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:9]))"; }
B="python bench.py --steps 60 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
for t in 20 21 2e 0f 2f; do
	$B --types $t 2>/dev/null | line T_$t >> gpurun_out/s42/types.txt
done
