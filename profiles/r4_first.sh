#!/bin/bash
# round 4, first GPU call: the new parity tests, the driver's bench command, a machine question (mixed VALU/SALU issue),
# the window-length statistics of the bench batch, and an A/B of the biquad segment length
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
grep -E "MemTotal|MemAvailable" /proc/meminfo > $O/r4_meminfo.txt; nproc >> $O/r4_meminfo.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4_pytest.log 2>&1; echo "pytest rc $?" >> $O/r4_pytest.log
tail -5 $O/r4_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_bench20.json 2> $O/r4_bench20.err; tail -c 600 $O/r4_bench20.err
cut -c1-300 $O/r4_bench20.json
timeout 120 profiles/ubench/mixed_issue > $O/r4_mixed_issue.jsonl 2>&1; cat $O/r4_mixed_issue.jsonl
TFREC_AMD_DEBUG_WINHIST=1 timeout 300 python bench.py --steps 4 --warmup 1 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs 2> $O/r4_winhist.txt > /dev/null; grep WINHIST $O/r4_winhist.txt
timeout 900 bash profiles/ab.sh 3 > $O/r4_ab_seg.txt 2>&1; cat $O/r4_ab_seg.txt
