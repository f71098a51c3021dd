#!/bin/bash
# session 9: the driver's command line: does the host (OpenMP threads of the parity gate spinning after it, cgroup CPU quota) cost the timed region?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
cat /sys/fs/cgroup/cpu.max > gpurun_out/s9/cgroup.txt 2>&1; nproc >> gpurun_out/s9/cgroup.txt; cat /sys/fs/cgroup/cpu.stat >> gpurun_out/s9/cgroup.txt 2>&1
run() {
	label=$1; shift
	for i in 1 2 3; do
		env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs --h2d-steps 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %7.3f ms/step steady %s  first %.1f  max %.1f  %s' % ('$label', j['ms_per_step'], j.get('ms_per_step_steady'), j['step_ms'][0], max(j['step_ms'][1:]), j['step_ms']))
" >> gpurun_out/s9/driver.txt 2>&1
	done
}
run default X=1
run omp_passive OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0
run omp_passive_t8 OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0 OMP_NUM_THREADS=8
cat /sys/fs/cgroup/cpu.stat >> gpurun_out/s9/cgroup.txt 2>&1
