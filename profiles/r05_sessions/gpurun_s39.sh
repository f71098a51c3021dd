#!/bin/bash
# session 39: 600 more campaign rounds on the round's final tree (HEAD = the sources of commit 9e51a0d), driver command once more
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s39
for seed in 1501 1502 1503 1504 1505 1506 1507 1508 1509 1510; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s39/campaign.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s39/driver_line.json 2> gpurun_out/s39/driver_line.err
