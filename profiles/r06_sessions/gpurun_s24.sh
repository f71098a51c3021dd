#!/bin/bash
# round 6 session 24: the front end as 2048 persistent workgroups (now the default): full GPU suite, campaign; A/B against a workgroup per tile
# (TFREC_AMD_FE_PERSIST=0) and with the discriminator pass persistent too (TFREC_AMD_FMDEV_PERSIST=n); 100 steps, two rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s24
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 2 100 8 p2048=default p0=default,TFREC_AMD_FE_PERSIST=0 fm4096=default,TFREC_AMD_FMDEV_PERSIST=4096 fm2048=default,TFREC_AMD_FMDEV_PERSIST=2048 fm8192=default,TFREC_AMD_FMDEV_PERSIST=8192 > $O/ab.txt 2>&1
for seed in 7001 7002; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
