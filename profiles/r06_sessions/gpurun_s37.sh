#!/bin/bash
# (bench.py carried a temporary --gc-off switch for this session only: the collector never ran inside a timed region, the switch was removed again)
# round 6 session 37: the slow runs of the driver's line are HOST stalls (session 36's trace: the GPU's copy of batch 8 done at 54.6 ms, the host's drain returns at 64.1):
# the driver's full command with the host's side timed (host_ms: longest submit, longest gap between drains, garbage collector pauses), 9 times as is and 9 times with
# Python's cyclic collector off inside the warm-up + timed region, alternating
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s37
mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9; do
	for g in 0 1; do
		python bench.py --gpus 1 --steps 20 --warmup 5 --gc-off $g 2>/dev/null | tail -1 > $O/line.json
		python -c "
import json; j=json.loads(open('$O/line.json').read()); print('gc_off=$g', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'max step', max(j['step_ms'][1:]), j['host_ms'])" >> $O/runs.txt
	done
done
cat $O/runs.txt | cut -c1-260
exit 0
