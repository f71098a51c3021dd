#!/bin/bash
# what each kernel group costs the batch period TODAY (TFREC_AMD_SKIP leaves kernels out: results wrong, timing only),
# after a parity run of the merged drain copy
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "fifo or drain or overflow or steady or bench_single" > $O/r4_pytest2.log 2>&1; tail -3 $O/r4_pytest2.log
run() {
	env $1 python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --steps 40 --warmup 14 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:8]
print('%-28s %7.3f ms/step steady %.3f (min %.2f med %.2f)  %s' % ('$1', j['ms_per_step'], j['ms_per_step_steady'] or 0, j['ms_min'], j['ms_median'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
}
for cfg in TFREC_AMD_SKIP=0 TFREC_AMD_SKIP=512 TFREC_AMD_SKIP=16 TFREC_AMD_SKIP=48 TFREC_AMD_SKIP=5 TFREC_AMD_SKIP=10 TFREC_AMD_SKIP=64 TFREC_AMD_SKIP=128 TFREC_AMD_SKIP=256 TFREC_AMD_SKIP=0; do
	run $cfg
done > $O/r4_whatif.txt 2>&1
cat $O/r4_whatif.txt
