#!/bin/bash
# round 6 session 16 (session 12 again, without the per-group atomics that masked it): is whb_demod_kernel<false> waiting for memory?  Steps of stage-1 outputs held ahead (TFREC_AMD_WHB_AHEAD = 1 (default), 2, 3);
# 100 steps, two rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s16
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 100 8 ah1=ah1 ah2=ah2 ah3=ah3 > $O/ab.txt 2>&1
exit 0
