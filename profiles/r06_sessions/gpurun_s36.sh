#!/bin/bash
# round 6 session 36: looking for the slow state of the driver's line (6.2 ms in session 34, 5.97 in session 32's first run: both the first bench after a GPU suite):
# (GPU suite, then the driver's full command under the kernel tracer) x 4, then the full command 6 more times
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
O=$R/gpurun_out/s36
mkdir -p $O
one() {
	rocprofv3 --kernel-trace --output-format csv -d $O/trace$1 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$1.json 2> /dev/null
	f=$(find $O/trace$1 -name "*kernel_trace.csv" | head -1)
	cp "$f" $O/kernel_trace$1.csv; rm -rf $O/trace$1
	python -c "
import json; j=json.loads(open('$O/bench$1.json').read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j['step_ms'])" >> $O/runs.txt
}
for i in 1 2 3 4; do
	(cd $R && timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)
	one s$i
done
for i in 1 2 3 4 5 6; do one b$i; done
cat $O/runs.txt | cut -c1-220
exit 0
