#!/bin/bash
# session 7: clocks and power under the pipeline's load
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
rocm-smi --showclocks --showpower --showtemp > gpurun_out/s7/idle.txt 2>&1
( for i in $(seq 1 200); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|socclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/s7/smi_during.txt &
SMI=$!
python bench.py --steps 600 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --parity-after-streams 8 --no-extra-configs > gpurun_out/s7/bench600.json 2> gpurun_out/s7/bench600.err
kill $SMI 2>/dev/null
rocm-smi --showclocks --showpower > gpurun_out/s7/after.txt 2>&1
rocm-smi --showperflevel --showpowerplay --showmaxpower > gpurun_out/s7/caps.txt 2>&1 || true
