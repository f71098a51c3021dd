#!/bin/bash
# round 6 session 52: the profile set of profiles/run_round.sh for HEAD (default bench line incl. all legs, 200-step run, kernel trace + stats of the same command, steady-state
# steps, PMC passes, HBM traffic, VALU mix + measured issue costs)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s52
hipcc --offload-arch=gfx950 -O3 profiles/ubench/valu_issue.hip -o profiles/ubench/valu_issue > /dev/null 2>&1
bash profiles/run_round.sh s52/r06_final > gpurun_out/s52/run_round.log 2>&1
tail -3 gpurun_out/s52/run_round.log
ls gpurun_out/s52
exit 0
