#!/bin/bash
# session 11: wave priorities (s_setprio) of the front end / the latency-bound kernels / the biquad passes, at 1024 x 48 and at 2048 x 48
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
python profiles/ab_run.py gpurun_out/s11/ab_prio.jsonl 2 60 8 base=default fe1=fe1 fe2=fe2 fe3=fe3 lat1=lat1 spec1=spec1 fe2lat1=fe2lat1 > gpurun_out/s11/ab_prio.txt 2>&1
AB_BENCH_ARGS="--streams 2048 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs" python profiles/ab_run.py gpurun_out/s11/ab_prio2048.jsonl 1 30 6 base=default fe2=fe2 fe3=fe3 > gpurun_out/s11/ab_prio2048.txt 2>&1
