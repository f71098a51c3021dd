#!/bin/bash
# round 6 session 32: the tree after chains2.hip was cut into stage headers (same device code): the GPU suite, the driver's line three times,
# the default bench with its rocprof summary, 240 campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s32
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/driver_lines.jsonl; done
python bench.py > $O/default_bench.json 2> $O/default_bench.err
for seed in 7301 7302 7303 7304; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/campaign.txt; done
python - <<'P' > $O/summary.txt
import json
for l in open('gpurun_out/s32/driver_lines.jsonl'):
    j=json.loads(l); print('driver line', j['ms_per_step'], j['value'], j['config'].get('parity_ok'), j.get('parity_after_timed'))
j=json.loads(open('gpurun_out/s32/default_bench.json').read().strip().splitlines()[-1])
print('default', j['steps'], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('whole_path_frac'))
P
cat $O/summary.txt $O/campaign.txt
exit 0
