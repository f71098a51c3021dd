#!/bin/bash
# usage (on the GPU box): profiles/run_pmc.sh <tag> "<counters>"  -> gpurun_out/<tag>_pmc.txt  (PMC pass only: no trace domains)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmc}
ctr=${2:-"SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"}
cd /tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$tag -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/pmc_$tag.log 2>&1
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$tag/*/*counter_collection.csv | head -1)) > $R/gpurun_out/${tag}_pmc.txt
cat $R/gpurun_out/${tag}_pmc.txt
