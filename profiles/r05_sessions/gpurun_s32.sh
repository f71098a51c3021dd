#!/bin/bash
# session 32: with cheap cooperative slicers, where should a window count as "long"?  TFREC_AMD_COOP_MIN 4096 (default) / 2048 / 1024 / 512 / 356
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s32
python profiles/ab_run.py gpurun_out/s32/ab.jsonl 2 100 8 c4096=default c2048=default,TFREC_AMD_COOP_MIN=2048 c1024=default,TFREC_AMD_COOP_MIN=1024 c512=default,TFREC_AMD_COOP_MIN=512 c356=default,TFREC_AMD_COOP_MIN=356 > gpurun_out/s32/ab.txt 2>&1
