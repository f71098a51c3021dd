#!/bin/bash
# round 6 session 35: the driver's 20-step command under the kernel tracer, 14 times: one run in eight lands in a slow state (6.2 ms against 5.75: batches finish in bursts
# of four) -- keep every run's trace to compare a slow run's timeline with a typical one
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
O=$R/gpurun_out/s35
mkdir -p $O
for i in $(seq 1 14); do
	rocprofv3 --kernel-trace --output-format csv -d $O/trace$i -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 > $O/bench$i.json 2> /dev/null
	f=$(find $O/trace$i -name "*kernel_trace.csv" | head -1)
	cp "$f" $O/kernel_trace$i.csv; rm -rf $O/trace$i
	python -c "
import json; j=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print($i, j['ms_per_step'], j['step_ms'])" >> $O/runs.txt
done
cat $O/runs.txt | cut -c1-200
exit 0
