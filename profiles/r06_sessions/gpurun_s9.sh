#!/bin/bash
# round 6 session 9: which of the round's structural changes to keep, one box, 100 steps, three rounds alternating (all without whb_zsr_kernel,
# session 8: -20 %): rows = the check a stream per row of 16 lanes (rounds 3-5), lanes = a stream per lane (whb_check.h);
# spec0 = the TFA_2 family's speculative biquad pass at the head of k2 (rounds 1-5), spec1 = on its own stream
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s9
mkdir -p $O
Z=TFREC_AMD_WHB_ZSR=0
python profiles/ab_run.py $O/ab.jsonl 3 100 8 rows_spec0=default,$Z,TFREC_AMD_WHB_CHECK_ROWS=1,TFREC_AMD_SPEC_OWN=0 lanes_spec0=default,$Z,TFREC_AMD_SPEC_OWN=0 \
	rows_spec1=default,$Z,TFREC_AMD_WHB_CHECK_ROWS=1 lanes_spec1=default,$Z > $O/ab.txt 2>&1
exit 0
