#!/bin/bash
# session 43: the speculative biquad pass on its own stream (spec.so) for the workloads WITHOUT the WHB chain, where k2 alone sets the period
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s43
for t in 07 0f 2f; do
	echo "== -T $t" >> gpurun_out/s43/ab.txt
	AB_BENCH_ARGS="--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs --types $t" python profiles/ab_run.py gpurun_out/s43/ab_$t.jsonl 2 80 8 spec=spec before=default >> gpurun_out/s43/ab.txt 2>&1
done
