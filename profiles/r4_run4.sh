#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
bash profiles/ab.sh 2 > $O/r4_ab4.txt 2>&1; cat $O/r4_ab4.txt
