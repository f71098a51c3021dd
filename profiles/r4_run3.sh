#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_pytest3.log 2>&1; tail -3 $O/r4_pytest3.log
bash profiles/ab_env.sh 2 "TFREC_AMD_REPAIR_KW=0" "TFREC_AMD_REPAIR_KW=1 TFREC_AMD_SCAN_KW=0" "TFREC_AMD_SCAN_KW=0" > $O/r4_ab_repairkw.txt 2>&1; cat $O/r4_ab_repairkw.txt
