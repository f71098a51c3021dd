#!/bin/bash
# round 6 session 29: session 28 again (its box stuttered: steps of 0.6 ms beside steps of 20): the front end as 2048 (default) / 3072 / 4096 / 8192
# persistent workgroups; 100 steps, three rounds; ms_per_step and the median step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s29
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 3 100 8 default=default fe3072=default,TFREC_AMD_FE_PERSIST=3072 fe4096=default,TFREC_AMD_FE_PERSIST=4096 fe8192=default,TFREC_AMD_FE_PERSIST=8192 > $O/ab.txt 2>&1
exit 0
