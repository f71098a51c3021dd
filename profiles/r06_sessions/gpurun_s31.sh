#!/bin/bash
# round 6 session 31: the one-stream configurations (BASELINE configs[0], [1]) are bound by the host's time per submit: deep layout (ten streams, default)
# against the shallow one (TFREC_AMD_DEEP=0: five streams, fewer events), and the host's time per submit (TFREC_AMD_HOST_PROF=1)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s31
mkdir -p $O
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], j['config']['parity_ok'], j['value'])"; }
B="python bench.py --experiments --streams 1 --blocks 48 --steps 300 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 1 --no-extra-configs"
for rep in 1 2; do
TFREC_AMD_DEEP=1 $B --types 7 2>/dev/null | line T7_deep >> $O/small.txt
TFREC_AMD_DEEP=0 $B --types 7 2>/dev/null | line T7_shallow >> $O/small.txt
TFREC_AMD_DEEP=1 $B --types 1 --thresh 0 2>/dev/null | line T1_auto_deep >> $O/small.txt
TFREC_AMD_DEEP=0 $B --types 1 --thresh 0 2>/dev/null | line T1_auto_shallow >> $O/small.txt
TFREC_AMD_DEEP=1 $B --types 2f 2>/dev/null | line T2f_deep >> $O/small.txt
TFREC_AMD_DEEP=0 $B --types 2f 2>/dev/null | line T2f_shallow >> $O/small.txt
done
TFREC_AMD_HOST_PROF=1 $B --types 7 --steps 40 2>&1 | grep -i "host\|prof" | tail -5 > $O/host_prof.txt
exit 0
