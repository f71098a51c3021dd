#!/bin/bash
# round 6 session 27: full-size parity on the final tree (1024 x 48 x 5 protocols, every flush of every stream, windows cut by a second submit) on other data:
# four seeds at the bench's noise level, two at higher noise (more glitches for the TFA_2 walk, denser candidates for TFA_1)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s27
timeout 1500 python profiles/ubench/fullsize_check.py 4001 4002 4003 4004:768 4005:1536 > gpurun_out/s27/fullsize.txt 2>&1
