#!/bin/bash
# round 6 session 11: the power sum of a locked window at its end instead of per step (whb_demod_kernel), the statistics counters split,
# new tests (config 5 at 64 streams x 48 blocks, the lane-per-step slicers' counters): GPU suite, A/B against the commit before, campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s11
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "config5 or cooperative or whb or steady" 2>&1 | tail -25 > $O/pytest_new.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 new=default prev=prev > $O/ab.txt 2>&1
for seed in 6501 6502; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
