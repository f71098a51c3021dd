#!/bin/bash
# session 46: full-size parity (1024 x 48 x 5 protocols, every flush of every stream, windows cut by a second submit) on other data:
# four seeds at the bench's noise level, two at higher noise (more glitches for the TFA_2 walk, denser candidates for TFA_1)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s46
timeout 1500 python profiles/ubench/fullsize_check.py 3001 3002 3003 3004 3005:768 3006:1536 > gpurun_out/s46/fullsize.txt 2>&1
