#!/bin/bash
# round 6 session 23 (session 22 again with the front end at its 97 registers: there the loop had cost it 56): the front end as n persistent workgroups (session 21: -6 % over 60 steps): 100 steps, three rounds; front-end tests with it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s23
mkdir -p $O
TFREC_AMD_FE_PERSIST=2048 timeout 900 python -m pytest tests -m gpu -x -q -k "frontend or wide or config5 or steady or hostile" 2>&1 | tail -4 > $O/pytest_fe.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 default=default p2048=default,TFREC_AMD_FE_PERSIST=2048 p3072=default,TFREC_AMD_FE_PERSIST=3072 p1536=default,TFREC_AMD_FE_PERSIST=1536 > $O/ab.txt 2>&1
exit 0
