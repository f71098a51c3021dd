#!/bin/bash
# session 40: whb_demod_kernel: three steps' loads in flight by renaming (loop body three times) instead of register moves (the move of
# the value requested in the same iteration was a vmcnt(0) at every loop end): WHB tests, A/B against the tree before (final.so), GPU suite, campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s40
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady_state or config2_full_size or bits_mode or state_carries" 2>&1 | tail -5 > gpurun_out/s40/pytest_some.txt
python profiles/ab_run.py gpurun_out/s40/ab.jsonl 3 100 8 new=default before=final > gpurun_out/s40/ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/s40/pytest.txt
for seed in 1601 1602 1603; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s40/campaign.txt; done
