#!/bin/bash
# round 6 session 13: the slicers' statistics counted per wave (head2) against the commit before the counters (prev), and the power sum at the
# window's end on top of it (psum: session 11's A/B of it was masked by the per-group atomics); 100 steps, three rounds; quick parity of head2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s13
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "cooperative or steady" 2>&1 | tail -5 > $O/pytest_new.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 head2=head2 prev=prev psum=psum > $O/ab.txt 2>&1
exit 0
