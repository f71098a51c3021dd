#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/ab_env.sh 2 "TFREC_AMD_FMDEV_OWN=2" "TFREC_AMD_FMDEV_OWN=2 TFREC_AMD_PRIO=hlhhhnn" "TFREC_AMD_FMDEV_OWN=1 TFREC_AMD_PRIO=hlhhhnn" "TFREC_AMD_FMDEV_OWN=2 TFREC_AMD_PRIO=hnhhhnl" "TFREC_AMD_FMDEV_OWN=2 TFREC_AMD_SCAN_KW=0" > gpurun_out/r4_ab6b.txt 2>&1; cat gpurun_out/r4_ab6b.txt
TFREC_AMD_FMDEV_OWN=2 bash profiles/run_steps.sh r4_fmown > /dev/null 2>&1
