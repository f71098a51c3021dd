#!/bin/bash
# session 34: the tree with the lane-per-step cooperative slicers and long windows from 1024 samples: GPU suite, profile set
# (profiles/run_round.sh r05_final), 1000-step soak with the parity check after it, driver line x3, 360 campaign rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s34
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/s34/pytest.txt
bash profiles/run_round.sh r05_final > gpurun_out/s34/run_round.log 2>&1
python bench.py --steps 1000 --warmup 8 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-after-streams 64 > gpurun_out/s34/soak_1000steps.json 2> gpurun_out/s34/soak.err
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s34/driver_line_$i.json 2>/dev/null; done
for seed in 1301 1302 1303 1304 1305 1306; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s34/campaign.txt; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s34/smoke.txt 2>&1
