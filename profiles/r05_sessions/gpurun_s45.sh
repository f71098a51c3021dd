#!/bin/bash
# session 45: FIFO depth 5 / 6 again (libraries built with TFREC_AMD_FIFO_DEPTH = 5 / 6: fifo5.so / fifo6.so) on the final tree: depth 2 -> 3 -> 4 is 7.4 -> 6.05 -> 5.42 ms
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s45
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['step_ms'] or []; print('%-14s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], 'first', (s[0] if s else None), j['config']['parity_ok'], j['config']['parity_after_timed'], j['roofline']['context_memory']['device_bytes'])"; }
A="--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
for rep in 1 2; do
	python bench.py $A --steps 100 --warmup 8 2>/dev/null | line d4_100 >> gpurun_out/s45/depth.txt
	python bench.py $A --steps 20 --warmup 5 2>/dev/null | line d4_20 >> gpurun_out/s45/depth.txt
	for d in 5 6; do
		TFREC_AMD_LIB=$R/tfrec_amd/ab/fifo$d.so PY_FIFO_DEPTH=$d python profiles/ubench/bench_depth.py $A --depth $d --steps 100 --warmup 10 2>gpurun_out/s45/err_$d.txt | line d${d}_100 >> gpurun_out/s45/depth.txt
		TFREC_AMD_LIB=$R/tfrec_amd/ab/fifo$d.so PY_FIFO_DEPTH=$d python profiles/ubench/bench_depth.py $A --depth $d --steps 20 --warmup 7 2>>gpurun_out/s45/err_$d.txt | line d${d}_20 >> gpurun_out/s45/depth.txt
	done
done
