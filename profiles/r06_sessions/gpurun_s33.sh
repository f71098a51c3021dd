#!/bin/bash
# round 6 session 33: where the FIRST batch of the driver's 20-step line spends its 19 ms (the pipeline is empty when the timed region starts):
# kernel trace of `bench.py --steps 20 --warmup 5`, every dispatch with its queue and its start / end
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
O=$R/gpurun_out/s33
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 > $O/bench.json 2> $O/bench.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/first_batch.txt <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
P
cp "$f" $O/kernel_trace.csv
rm -rf $O/trace
ls -la $O
exit 0
