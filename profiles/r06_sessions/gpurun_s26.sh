#!/bin/bash
# round 6 session 26: the final tree (front end and discriminator pass persistent) -- the cooperative slicers' grid (32768 one-wave workgroups on high-priority
# streams: TFREC_AMD_COOP_BLOCKS), 1000-step soak with the parity gate behind it, the period by protocol subset, 360 campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s26
mkdir -p $O
python profiles/ab_run.py $O/ab.jsonl 2 100 8 cb32768=default cb4096=default,TFREC_AMD_COOP_BLOCKS=4096 cb1024=default,TFREC_AMD_COOP_BLOCKS=1024 cb8192=default,TFREC_AMD_COOP_BLOCKS=8192 > $O/ab.txt 2>&1
python bench.py --steps 1000 --warmup 8 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-after-streams 64 > $O/soak_1000steps.json 2> $O/soak.err
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:9]))"; }
B="python bench.py --steps 60 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
for t in 20 21 2e 0f 2f; do
	$B --types $t 2>/dev/null | line T_$t >> $O/types.txt
done
python bench.py > $O/default_bench.json 2> $O/default_bench.err
for seed in 7201 7202 7203 7204 7205 7206; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
