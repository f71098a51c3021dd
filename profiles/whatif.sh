#!/bin/bash
# What each kernel group costs the batch period (TFREC_AMD_SKIP leaves kernels out: results are WRONG, timing only).
# usage (GPU box, repo root): profiles/whatif.sh [skip masks ...]   (default: the set of profiles/r04_whatif.txt)
#   bits: 1 TFA_2 coop slicers, 2 TFA_1 coop slicers, 4 TFA_2 slicers, 8 TFA_1 mark + slicers, 16 WHB check, 32 whb_demod,
#         64 discriminator, 128 TFA_2 biquads, 256 WHB biquads, 512 front end (after the twelfth submit)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
masks=${@:-0 512 16 48 5 10 128 256 0}
for m in $masks; do
	TFREC_AMD_SKIP=$m python bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --steps 40 --warmup 14 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:8]
print('TFREC_AMD_SKIP=%-4s %7.3f ms/step steady %.3f (min %.2f med %.2f)  %s' % ('$m', j['ms_per_step'], j['ms_per_step_steady'] or 0, j['ms_min'], j['ms_median'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
"
done
