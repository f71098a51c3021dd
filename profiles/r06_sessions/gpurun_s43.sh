#!/bin/bash
# round 6 session 43: drain_copy_kernel against hipMemcpyAsync over 200 steps (the steady period must not pay for the 20-step line), 4 alternating rounds,
# and the per-step times of two 20-step lines each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s43
mkdir -p $O
B="python bench.py --gpus 1 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 --experiments"
one() { # label, env, args
	env $2 $B $3 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); s=j['step_ms']; print('%-16s'%'$1', j['steps'], j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'max submit', j['host_ms']['submit_max'], j['config']['parity_ok'], j['config']['parity_after_timed'], s if s and len(s)<=20 else '')" >> $O/runs.txt
}
for i in 1 2 3 4; do
	one "copy kernel" "TFREC_AMD_COPY_KERNEL=1" "--steps 200 --warmup 8"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0" "--steps 200 --warmup 8"
done
for i in 1 2; do
	one "copy kernel" "TFREC_AMD_COPY_KERNEL=1" "--steps 20 --warmup 5"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0" "--steps 20 --warmup 5"
done
cat $O/runs.txt | cut -c1-300
exit 0
