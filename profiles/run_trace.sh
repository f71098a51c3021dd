#!/bin/bash
# usage (on the GPU box, from the repo root): profiles/run_trace.sh <tag>   -> gpurun_out/<tag>_{stats,timeline}.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-trace}
cd /tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --parity-streams 0 > $R/gpurun_out/prof_$tag.log 2>&1
for f in $R/gpurun_out/prof_$tag/*/*.db; do
	python $R/profiles/rocpd_summary.py $f > $R/gpurun_out/${tag}_stats.txt
	python $R/profiles/rocpd_timeline.py $f > $R/gpurun_out/${tag}_timeline.txt
done
rm -rf $R/gpurun_out/prof_$tag
cat $R/gpurun_out/${tag}_timeline.txt
