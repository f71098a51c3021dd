#!/bin/bash
# session 22: final tree: profile set, 1000-step soak with the parity check after it, 360 campaign rounds, driver line x3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s22
bash profiles/run_round.sh r05_final > gpurun_out/s22/run_round.log 2>&1
python bench.py --steps 1000 --warmup 8 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-after-streams 64 > gpurun_out/s22/soak_1000steps.json 2> gpurun_out/s22/soak.err
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s22/driver_line_$i.json 2>/dev/null; done
for seed in 901 902 903 904 905 906; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s22/campaign.txt; done
