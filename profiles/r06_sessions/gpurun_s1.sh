#!/bin/bash
# round 6 session 1: the tree of commit 576301a (product library without knobs, rotating batches) on the GPU:
# full GPU suite, the driver's command, a 200-step run, the kernel trace of a 12-step run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/s1/smoke.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s1/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/driver_line.json 2> gpurun_out/s1/driver_line.err
python bench.py --steps 200 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > gpurun_out/s1/bench200.json 2> gpurun_out/s1/bench200.err
cd /tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 12 --warmup 4 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
cp $(ls /tmp/kt/*/*kernel_stats.csv | head -1) $R/gpurun_out/s1/kernel_stats.csv
strings $R/tfrec_amd/libtfrec_amd.so | grep -c TFREC_AMD_ > $R/gpurun_out/s1/knob_strings.txt
