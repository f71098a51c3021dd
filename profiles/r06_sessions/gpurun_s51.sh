#!/bin/bash
# round 6 session 51: HEAD (the register ring in whb_demod_kernel, the copy kernel): 180 campaign rounds, the driver's command twice, the 200-step run, the tail case
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s51
mkdir -p $O
for seed in 7701 7702 7703; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/out.txt; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/line$i.json; python -c "
import json; j=json.loads(open('$O/line$i.json').read()); print('driver line', j['ms_per_step'], j['value'], j['config']['parity_ok'], j['config']['parity_after_timed'], 'longest submit', j['host_ms']['submit_max'], 'other', [(round(v.get('ms_per_step',0),2)) for v in (j['other_configs'].values() if isinstance(j['other_configs'],dict) else j['other_configs'])])" >> $O/out.txt; done
python bench.py --steps 200 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 64 2>/dev/null | tail -1 > $O/bench_200steps.json
python -c "
import json; j=json.loads(open('$O/bench_200steps.json').read()); print('200 steps', j['ms_per_step'], j['value'], j['config']['parity_ok'], j['config']['parity_after_timed'], {k:round(v,2) for k,v in j['roofline']['kernels_ms'].items() if v>2})" >> $O/out.txt
cat $O/out.txt
exit 0
