#!/bin/bash
# session 19: biquad passes with the next window's descriptor read ahead: parity, A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s19
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s19/pytest.txt
python profiles/ab_run.py gpurun_out/s19/ab.jsonl 3 60 8 new=default old=prev > gpurun_out/s19/ab.txt 2>&1
