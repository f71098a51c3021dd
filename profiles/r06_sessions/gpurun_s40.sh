#!/bin/bash
# round 6 session 40: which HIP runtime limit makes submits 5-7 after a synchronize block for a batch period each?  (session 39: not the signal pool)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s40
mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8"
one() { # label, env
	env $2 $B 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); print('%-34s'%'$1', j['ms_per_step'], 'submits', j['host_ms']['submit_ms'])" >> $O/runs.txt
}
one "default" "X=1"
one "kernarg pool 16 MB" "HSA_KERNARG_POOL_SIZE=16777216"
one "batch cpu sync 100000" "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=100000"
one "max batch 100000" "DEBUG_CLR_MAX_BATCH_SIZE=100000"
one "max batch 8" "DEBUG_CLR_MAX_BATCH_SIZE=8"
one "command buffers 64" "GPU_MAX_COMMAND_BUFFERS=64"
one "aql queue 65536" "ROC_AQL_QUEUE_SIZE=65536"
one "cpu wait for signal 0" "ROC_CPU_WAIT_FOR_SIGNAL=0"
one "active wait timeout 0" "ROC_ACTIVE_WAIT_TIMEOUT=0"
one "default" "X=1"
cat $O/runs.txt | cut -c1-260
exit 0
