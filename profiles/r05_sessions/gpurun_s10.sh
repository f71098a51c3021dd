#!/bin/bash
# session 10: device-wide activity accumulators (gfx / memory controller) under the pipeline's load
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
python bench.py --steps 2500 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --parity-after-streams 8 --no-extra-configs > gpurun_out/s10/bench.json 2>/dev/null &
B=$!
for i in $(seq 1 40); do
	echo "t=$(date +%s.%N) $(rocm-smi --showmetrics 2>/dev/null | grep -E 'average_gfx_activity|average_umc_activity|gfx_activity_acc|mem_activity_acc|accumulation_counter|current_gfxclk |xcp_stats.gfx_busy_acc' | sed 's/GPU\[0\]//' | tr -s '\t ' ' ' | tr '\n' ';')" >> gpurun_out/s10/acc.txt
	kill -0 $B 2>/dev/null || break
	sleep 0.7
done
wait $B
