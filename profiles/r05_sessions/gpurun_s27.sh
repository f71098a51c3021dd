#!/bin/bash
# session 27: is the batch sensitive to instruction-cache pressure?  bench.py (100 steps) alone, beside a 2 KB-loop polluter
# (control: same instructions, same rate) and beside 24 KB / 48 KB-loop polluters (profiles/ubench/ipollute.hip, own process)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s27
B="python bench.py --steps 100 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:8]))"; }
for round in 1 2; do
	$B 2>/dev/null | line alone >> gpurun_out/s27/pollute.txt
	for kb in 2 24 48; do
		profiles/ubench/ipollute $kb 45 > /tmp/pol_$kb.txt 2>&1 &
		PID=$!
		sleep 2
		$B 2>/dev/null | line beside_${kb}KB >> gpurun_out/s27/pollute.txt
		kill $PID 2>/dev/null; wait $PID 2>/dev/null
	done
done
