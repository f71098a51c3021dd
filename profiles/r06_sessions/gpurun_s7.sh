#!/bin/bash
# round 6 session 7: whb_chain_kernel variants, 40 steps each: double2 LDS reads (lanes2), the consumer alone on its SIMD (claim),
# 16 streams per workgroup (s16), the chain without compare / add-with-carry (nocmp: timing only, parity off)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s7
mkdir -p $O
AB_BENCH_ARGS="--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs" python profiles/ab_run.py $O/ab.jsonl 2 40 8 lanes2=lanes2 claim=claim s16=s16 rows=lanes2,TFREC_AMD_WHB_CHECK_ROWS=1 > $O/ab.txt 2>&1
AB_BENCH_ARGS="--cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs" python profiles/ab_run.py $O/ab_nocmp.jsonl 2 40 8 nocmp=nocmp > $O/ab_nocmp.txt 2>&1
exit 0
