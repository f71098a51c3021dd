#!/bin/bash
# round 6 session 10: whb_chain_kernel -- consumer wave priority 3, producer loads three rounds ahead, both; 100 steps, two rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s10
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 100 8 ctl=half prio3=prio3 pf3=pf3 prio3pf3=prio3pf3 > $O/ab.txt 2>&1
exit 0
