#!/bin/bash
# session 17: the final tree: GPU suite, the driver's command, 300 more campaign rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s17
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s17/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s17/driver_line.json 2> gpurun_out/s17/driver_line.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s17/smoke.txt 2>&1
for seed in 801 802 803 804 805; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s17/campaign.txt; done
