#!/bin/bash
# session 12: GPU suite, the round's profile set (profiles/run_round.sh), randomised campaign
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/s12/pytest.txt
bash profiles/run_round.sh r05_mid > gpurun_out/s12/run_round.log 2>&1
for seed in 501 502 503 504 505; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s12/campaign.txt; done
