#!/bin/bash
# round 6 session 42: the drain's copy as a kernel of our own (drain_copy_kernel) against hipMemcpyAsync (TFREC_AMD_COPY_KERNEL=0, experiments build): the host's time in every
# submit of the driver's 20-step line, 5 alternating rounds; then the GPU suite on the new tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s42
mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 --experiments"
one() { # label, env
	env $2 $B 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); print('%-16s'%'$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], j['config']['parity_ok'], j['config']['parity_after_timed'], 'submits', j['host_ms']['submit_ms'])" >> $O/runs.txt
}
for i in 1 2 3 4 5; do
	one "copy kernel" "TFREC_AMD_COPY_KERNEL=1"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0"
done
cat $O/runs.txt | cut -c1-250
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
exit 0
