#!/bin/bash
# session 2: stream groups (tests + A/B on the driver's command line), bench legs, seg256 divisors
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_groups or steady_state" 2>&1 | tail -15 > gpurun_out/s2/pytest.txt
# driver's command (20 steps, 5 warm-up): groups 1 / 2 / 4 / 8, base and seg256
AB_BENCH_ARGS="--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs" python profiles/ab_run.py gpurun_out/s2/ab_groups.jsonl 3 20 5 \
  g1=default g2=default,TFREC_AMD_GROUPS=2 g4=default,TFREC_AMD_GROUPS=4 g8=default,TFREC_AMD_GROUPS=8 \
  s256g1=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6 \
  s256g4=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6,TFREC_AMD_GROUPS=4 \
  > gpurun_out/s2/ab_groups.txt 2>&1
# long runs: steady state
python profiles/ab_run.py gpurun_out/s2/ab_long.jsonl 1 100 8 \
  g1=default g2=default,TFREC_AMD_GROUPS=2 g4=default,TFREC_AMD_GROUPS=4 \
  s256g1=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6 \
  s256g2=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6,TFREC_AMD_GROUPS=2 \
  s256d3=seg256,TFREC_AMD_SPEC_DIV=3,TFREC_AMD_REPAIR_DIV=6 \
  s256d6=seg256,TFREC_AMD_SPEC_DIV=6,TFREC_AMD_REPAIR_DIV=8 \
  s256q8=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6,GPU_MAX_HW_QUEUES=8 \
  > gpurun_out/s2/ab_long.txt 2>&1
# the full line with the new legs
python bench.py --steps 20 --warmup 5 > gpurun_out/s2/driver_line.json 2> gpurun_out/s2/driver_line.err
