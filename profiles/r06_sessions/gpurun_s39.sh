#!/bin/bash
# round 6 session 39: the two 6 ms submits of every 20-step line (host_ms.submit_ms of session 38: always the 6th and 7th of the timed region = the 12th and 13th of the context):
# do they follow the context's submit count (warm-up 2 / 8), and the HIP runtime's signal pool (ROC_SIGNAL_POOL_SIZE)?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s39
mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8"
one() { # label, env, args
	env $2 $B $3 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); print('%-28s'%'$1', j['ms_per_step'], 'submits', j['host_ms']['submit_ms'])" >> $O/runs.txt
}
one "warmup5" "X=1" "--warmup 5"
one "warmup5" "X=1" "--warmup 5"
one "warmup2" "X=1" "--warmup 2"
one "warmup8" "X=1" "--warmup 8"
one "pool256 warmup5" "ROC_SIGNAL_POOL_SIZE=256" "--warmup 5"
one "pool256 warmup5" "ROC_SIGNAL_POOL_SIZE=256" "--warmup 5"
one "pool1024 warmup5" "ROC_SIGNAL_POOL_SIZE=1024" "--warmup 5"
one "hwq8 warmup5" "GPU_MAX_HW_QUEUES=8" "--warmup 5"
cat $O/runs.txt | cut -c1-260
exit 0
