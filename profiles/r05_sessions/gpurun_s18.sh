#!/bin/bash
# session 18: the driver's command with caller-side pacing of the submits (de-bunching)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s18
for i in 1 2 3 4; do for pace in 0 0.7 0.85; do
	python bench.py --gpus 1 --steps 20 --warmup 5 --pace $pace --no-extra-configs --h2d-steps 0 --cpu-budget 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pace %-5s %7.3f ms/step steady %s  %s' % ('$pace', j['ms_per_step'], j.get('ms_per_step_steady'), j['step_ms']))
" >> gpurun_out/s18/pace.txt 2>&1
done; done
