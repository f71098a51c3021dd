#!/bin/bash
# usage (GPU box, repo root): [env ...] profiles/run_steps.sh <tag> [bench args]  -> gpurun_out/<tag>_steps.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-steps}
shift
cd /tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -- python $R/bench.py --steps 14 --warmup 3 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs "$@" > $R/gpurun_out/prof_$tag.log 2>&1
for f in $R/gpurun_out/prof_$tag/*/*.db; do
	python $R/profiles/rocpd_steps.py $f > $R/gpurun_out/${tag}_steps.txt
done
rm -rf $R/gpurun_out/prof_$tag
tail -3 $R/gpurun_out/${tag}_steps.txt
