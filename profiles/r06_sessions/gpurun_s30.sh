#!/bin/bash
# round 6 session 30: stream priorities again, now that the front end's queue empties at once (round 4: every assignment other than the default was worse
# -- with a front end that starved below high priority).  TFREC_AMD_PRIO = one letter per stream fs cp cs t1 aux k2 kw (default hnhhhnn); the discriminator's
# and the speculative biquad pass's own streams (default low) by TFREC_AMD_FMDEV_OWN / _SPEC_OWN = 2 (normal) / 3 (high); 100 steps, two rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s30
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 100 8 default=default allhigh=default,TFREC_AMD_PRIO=hhhhhhh allnorm=default,TFREC_AMD_PRIO=nnnnnnn \
	allhigh_x=default,TFREC_AMD_PRIO=hhhhhhh,TFREC_AMD_FMDEV_OWN=3,TFREC_AMD_SPEC_OWN=3 allnorm_x=default,TFREC_AMD_PRIO=nnnnnnn,TFREC_AMD_FMDEV_OWN=2,TFREC_AMD_SPEC_OWN=2 \
	fsnorm=default,TFREC_AMD_PRIO=nnhhhnn k2high=default,TFREC_AMD_PRIO=hnhhhhh > $O/ab.txt 2>&1
exit 0
