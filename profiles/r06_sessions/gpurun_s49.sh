#!/bin/bash
# round 6 session 49: whb_demod_kernel's step loop unrolled by four over a ring of four registers with fixed roles (no move of an in-flight register: three steps of
# stage-1 outputs really in flight) against the tree before (tfrec_amd/ab/old.so): 200 steps, 3 alternating rounds; the GPU suite on the new tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s49
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 3 200 8 ring=ring old=old > $O/ab.txt 2>&1
cat $O/ab.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
exit 0
