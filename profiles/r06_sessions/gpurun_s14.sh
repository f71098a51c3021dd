#!/bin/bash
# round 6 session 14: L2 atomics per kernel (after session 12: 180 k same-address atomics per batch cost 25 %): TCC_ATOMIC_sum, TCC_REQ_sum
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s14
mkdir -p $O
cd /tmp
rm -rf /tmp/pmc_at
timeout 600 rocprofv3 --pmc TCC_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_at -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2> $O/err.txt
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_at/*/*counter_collection.csv | head -1)) > $O/pmc_atomics.txt 2>> $O/err.txt
exit 0
