#!/bin/bash
# round 6 session 28: how many persistent workgroups (front end: 1024 = what a chip holds at 37 KB of LDS each, 1280, 2048 (default); discriminator pass: 2048,
# 4096 (default), 16384); 100 steps, two rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s28
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 100 8 default=default fe1024=default,TFREC_AMD_FE_PERSIST=1024 fe1280=default,TFREC_AMD_FE_PERSIST=1280 fe4096=default,TFREC_AMD_FE_PERSIST=4096 fm2048=default,TFREC_AMD_FMDEV_PERSIST=2048 fm16384=default,TFREC_AMD_FMDEV_PERSIST=16384 > $O/ab.txt 2>&1
exit 0
