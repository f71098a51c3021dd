#!/bin/bash
# round 6 session 19: the final tree -- 1000-step soak with the parity gate behind it, the period by protocol subset (-T 20 / 21 / 2e / 0f / 2f),
# what-if runs (experiments build: kernel groups left out, timing only), 360 more campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s19
mkdir -p $O
python bench.py --steps 1000 --warmup 8 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-after-streams 64 > $O/soak_1000steps.json 2> $O/soak.err
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:9]))"; }
B="python bench.py --steps 60 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
for t in 20 21 2e 0f 2f; do
	$B --types $t 2>/dev/null | line T_$t >> $O/types.txt
done
W="python bench.py --experiments --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --steps 60 --warmup 14 --no-extra-configs"
for m in 0 16 48 128 512 304 0; do
	TFREC_AMD_SKIP=$m $W 2>/dev/null | line SKIP_$m >> $O/whatif.txt
done
for seed in 6801 6802 6803 6804 6805 6806; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
