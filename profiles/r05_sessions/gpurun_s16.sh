#!/bin/bash
# session 16: the round's final profile set (profiles/run_round.sh r05_final) + campaign on the final build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s16
bash profiles/run_round.sh r05_final > gpurun_out/s16/run_round.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s16/driver_line.json 2> gpurun_out/s16/driver_line.err
for seed in 701 702 703 704; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s16/campaign.txt; done
