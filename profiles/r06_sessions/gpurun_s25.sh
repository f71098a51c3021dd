#!/bin/bash
# round 6 session 25: the final tree (front end and discriminator pass persistent) -- GPU suite, the profile set of profiles/run_round.sh (default bench line incl. all legs, 200-step run,
# kernel trace + stats, timeline, PMC passes, traffic, VALU mix), the driver's command three times, the first campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s25
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s25/pytest.txt
hipcc --offload-arch=gfx950 -O3 profiles/ubench/valu_issue.hip -o profiles/ubench/valu_issue > /dev/null 2>&1
bash profiles/run_round.sh s25/r06_final > gpurun_out/s25/run_round.log 2>&1
cd $R
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/s25/driver_line_$i.json; done
for seed in 7101 7102 7103; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s25/campaign.txt; done
exit 0
