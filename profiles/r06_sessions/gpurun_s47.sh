#!/bin/bash
# round 6 session 47: the final tree (the drain's copy = drain_copy_kernel) -- GPU suite, the profile set of profiles/run_round.sh (default bench line incl. all legs,
# 200-step run, kernel trace + stats, PMC passes, traffic, VALU roof), the driver's command three times, 240 campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s47
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -1 $O/pytest_gpu.txt
bash profiles/run_round.sh s47/r06_final > $O/run_round.log 2>&1
cd $R
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/driver_lines.jsonl; done
for seed in 7501 7502 7503 7504; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/campaign.txt; done
python - <<'P'
import json
for l in open('gpurun_out/s47/driver_lines.jsonl'):
    j=json.loads(l); print('driver line', j['ms_per_step'], j['value'], 'steady', j['ms_per_step_steady'], 'first', j['step_ms'][0], j['config']['parity_ok'], j['config']['parity_after_timed'], 'max submit', j['host_ms']['submit_max'])
for f in ('r06_final_bench.json','r06_final_bench_200steps.json'):
    j=json.loads(open('gpurun_out/s47/'+f).read().strip().splitlines()[-1]); r=j['roofline']
    print(f, j['steps'], j['ms_per_step'], j['value'], 'frac', r['frac'], r.get('whole_path_frac'), 'kernel ms', r['kernel_ms_hip_events'], r['kernel_ms_profiles'])
P
cat $O/campaign.txt
exit 0
