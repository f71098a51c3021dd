#!/bin/bash
# round 6 session 34: the final tree as the driver will run it: build() on the box, smoke(), the GPU suite three times over, the driver's bench command, 300 more campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s34
mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); import tfrec_amd._build as b; print('\n'.join(b.last_actions())); g.smoke(); print('smoke ok')" > $O/build_smoke.txt 2>&1
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 >> $O/pytest_gpu_x3.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
for seed in 7401 7402 7403 7404 7405; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/campaign.txt; done
tail -3 $O/build_smoke.txt; cat $O/pytest_gpu_x3.txt $O/campaign.txt
python -c "
import json; j=json.loads(open('$O/driver_line.json').read().strip().splitlines()[-1]); r=j['roofline']
print('driver line', j['ms_per_step'], j['value'], r['frac'], r['kernel_ms_hip_events'], r['kernel_ms_profiles'], r.get('kernel_ms_covers'), j['config'].get('parity_ok'), j['config'].get('parity_after_timed'))"
exit 0
