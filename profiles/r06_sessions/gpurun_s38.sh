#!/bin/bash
# round 6 session 38: which submits of the driver's line take milliseconds on the host (host_ms.submit_ms: every submit of the timed region), 6 full runs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s38
mkdir -p $O
for i in 1 2 3 4 5 6; do
	python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/line$i.json
	python -c "
import json; j=json.loads(open('$O/line$i.json').read()); print(j['ms_per_step'], 'steps', j['step_ms']); print('   submits', j['host_ms']['submit_ms'])" >> $O/runs.txt
done
cat $O/runs.txt | cut -c1-260
exit 0
