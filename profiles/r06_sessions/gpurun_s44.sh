#!/bin/bash
# round 6 session 44: drain_copy_kernel against hipMemcpyAsync over 200 steps, 6 rounds in A B B A order (the box drifts by 1-2 % over minutes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s44
mkdir -p $O
B="python bench.py --gpus 1 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 --experiments --steps 200 --warmup 8"
one() { # label, env
	env $2 $B 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); print('%-16s'%'$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'max submit', j['host_ms']['submit_max'], j['config']['parity_ok'], j['config']['parity_after_timed'])" >> $O/runs.txt
}
for i in 1 2 3; do
	one "copy kernel" "TFREC_AMD_COPY_KERNEL=1"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0"
	one "copy kernel" "TFREC_AMD_COPY_KERNEL=1"
done
cat $O/runs.txt
python - <<'P'
import re
a={'copy kernel':[], 'hipMemcpyAsync':[]}
for l in open('gpurun_out/s44/runs.txt'):
    k=l[:16].strip(); a[k].append(float(l[16:].split()[0]))
for k,v in a.items(): print(k, sum(v)/len(v), v)
P
exit 0
