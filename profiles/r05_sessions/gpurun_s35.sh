#!/bin/bash
# session 35: where whb_demod_kernel<false>'s cycles go (-DTFREC_AMD_PROFILE_WHB: cycle counters around the parts of a step), WHB alone and
# beside the other chains (all five protocols, submits in flight); 180 more campaign rounds on the final tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s35
for m in 20 2f; do
	for mode in alone pipelined; do
		echo "== types 0x$m $mode" >> gpurun_out/s35/whb_cycles.txt
		TFREC_AMD_LIB=$R/tfrec_amd/ab/whbprof.so python profiles/ubench/whb_cycles.py $m $mode >> gpurun_out/s35/whb_cycles.txt 2>&1
	done
done
for seed in 1307 1308 1309; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s35/campaign.txt; done
