#!/bin/bash
# round 6 session 46: drain_copy_kernel with 64 / 256 / 1024 workgroups over 200 steps (A B C C B A x 2), then hipMemcpyAsync beside them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s46
mkdir -p $O
B="python bench.py --gpus 1 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 --experiments --steps 200 --warmup 8"
one() { # label, env
	env $2 $B 2>/dev/null | tail -1 > $O/line.json
	python -c "
import json; j=json.loads(open('$O/line.json').read()); print('%-16s'%'$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'max submit', j['host_ms']['submit_max'], j['config']['parity_ok'], j['config']['parity_after_timed'])" >> $O/runs.txt
}
for i in 1 2; do
	one "blocks 64" "TFREC_AMD_COPY_BLOCKS=64"
	one "blocks 256" "TFREC_AMD_COPY_BLOCKS=256"
	one "blocks 1024" "TFREC_AMD_COPY_BLOCKS=1024"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0"
	one "hipMemcpyAsync" "TFREC_AMD_COPY_KERNEL=0"
	one "blocks 1024" "TFREC_AMD_COPY_BLOCKS=1024"
	one "blocks 256" "TFREC_AMD_COPY_BLOCKS=256"
	one "blocks 64" "TFREC_AMD_COPY_BLOCKS=64"
done
python - <<'P'
import collections
a=collections.defaultdict(list)
for l in open('gpurun_out/s46/runs.txt'):
    a[l[:16].strip()].append(float(l[16:].split()[0]))
for k,v in a.items(): print('%-16s %.4f'%(k, sum(v)/len(v)), v)
P
exit 0
