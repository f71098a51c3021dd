#!/bin/bash
# session 28: the batch cut in TIME (1024 streams x 24 / 16 / 12 / 8 blocks): period per 48 blocks and the cost of the pipeline's fill
# on a line of 20 x 48 blocks (the driver's command submits 20 batches behind an empty pipeline)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s28
B="python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['step_ms']; print('$1', 'ms/step', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'first', s[0], 'last2', s[-2:], j['config']['parity_ok'], j['config']['parity_after_timed'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:8]))"; }
for rep in 1 2; do
	$B --blocks 48 --steps 20 --warmup 5 2>/dev/null | line b48_20 >> gpurun_out/s28/shapes.txt
	$B --blocks 24 --steps 40 --warmup 10 2>/dev/null | line b24_40 >> gpurun_out/s28/shapes.txt
	$B --blocks 16 --steps 60 --warmup 15 2>/dev/null | line b16_60 >> gpurun_out/s28/shapes.txt
	$B --blocks 12 --steps 80 --warmup 20 2>/dev/null | line b12_80 >> gpurun_out/s28/shapes.txt
	$B --blocks 8 --steps 120 --warmup 30 2>/dev/null | line b8_120 >> gpurun_out/s28/shapes.txt
done
$B --blocks 24 --steps 400 --warmup 10 2>/dev/null | line b24_400 >> gpurun_out/s28/shapes.txt
$B --blocks 12 --steps 800 --warmup 20 2>/dev/null | line b12_800 >> gpurun_out/s28/shapes.txt
