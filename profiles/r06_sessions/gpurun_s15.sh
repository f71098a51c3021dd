#!/bin/bash
# round 6 session 15: biquad_repair_slots added once per wave instead of once per segment (head3 against head2b, both experiments builds: they also
# count the slicers' vector groups per wave); prod = the product library of head3 (no vector-group counters); 100 steps, three rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s15
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "steady or campaign" 2>&1 | tail -5 > $O/pytest_new.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 head3=head3 head2b=head2b prod=product > $O/ab.txt 2>&1
exit 0
