#!/bin/bash
# session 44: batches kept in flight by the caller (bench.py --depth 2 / 3 / 4): the driver's 20-step line and a 100-step run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s44
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['step_ms'] or []; print('%-14s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'first', (s[0] if s else None), j['config']['parity_ok'], j['config']['parity_after_timed'])"; }
B="python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
for rep in 1 2; do
for d in 4 3 2; do
	$B --depth $d --steps 20 --warmup 5 2>/dev/null | line d${d}_20 >> gpurun_out/s44/depth.txt
	$B --depth $d --steps 100 --warmup 8 2>/dev/null | line d${d}_100 >> gpurun_out/s44/depth.txt
done
done
