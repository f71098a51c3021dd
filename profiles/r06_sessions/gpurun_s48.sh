#!/bin/bash
# round 6 session 48: the tree as committed last (drain_copy_kernel in namespace tfrec): GPU suite, smoke(), the driver's command twice, 120 campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s48
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 > $O/out.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/out.txt
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/line$i.json; python -c "
import json; j=json.loads(open('$O/line$i.json').read()); print('driver line', j['ms_per_step'], j['value'], j['config']['parity_ok'], j['config']['parity_after_timed'], 'longest submit', j['host_ms']['submit_max'])" >> $O/out.txt; done
for seed in 7601 7602; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/out.txt; done
cat $O/out.txt
exit 0
