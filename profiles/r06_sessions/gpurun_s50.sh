#!/bin/bash
# round 6 session 50: with whb_demod_kernel shorter (session 49) whb_chain_kernel is the longest kernel of the batch alone: its variants again (consumer wave at priority 3,
# producers three rounds ahead, both, 16 streams per workgroup) -- nothing in sessions 7 / 10, when whb_demod_kernel was as long
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s50
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 200 8 ring=ring prio3=ring_prio3 pf3=ring_pf3 prio3pf3=ring_prio3pf3 s16=ring_s16 > $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-200
exit 0
