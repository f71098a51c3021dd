#!/bin/bash
# session 3: fused biquad launch (default) vs round-4 form; full GPU suite; ramp; stream splits; host time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s3/pytest.txt
python profiles/ab_run.py gpurun_out/s3/ab_fused.jsonl 2 60 8 \
  fused=default \
  nofuse=default,TFREC_AMD_FUSED_BIQUAD=0 \
  fused_norep2=default,TFREC_AMD_REPAIR2=0 \
  fused_t1early=default,TFREC_AMD_T1_EARLY=1 \
  fused_coop1=default,TFREC_AMD_COOP_STREAM=1 \
  fused_mark1=default,TFREC_AMD_MARK_OWN=1 \
  fused_both_q8=default,TFREC_AMD_COOP_STREAM=1,TFREC_AMD_MARK_OWN=1,GPU_MAX_HW_QUEUES=8 \
  fused_both=default,TFREC_AMD_COOP_STREAM=1,TFREC_AMD_MARK_OWN=1 \
  fused_s8=default,TFREC_AMD_SPEC_DIV=8 \
  fused_s3=default,TFREC_AMD_SPEC_DIV=3 \
  > gpurun_out/s3/ab_fused.txt 2>&1
python profiles/ab_run.py gpurun_out/s3/ab_ramp.jsonl 3 20 5 \
  r0=default r1=default,TFREC_BENCH_RAMP=1 r2=default,TFREC_BENCH_RAMP=2 r3=default,TFREC_BENCH_RAMP=3 \
  > gpurun_out/s3/ab_ramp.txt 2>&1
TFREC_AMD_HOST_PROF=1 python bench.py --thresh 30000 --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > gpurun_out/s3/never.json 2> gpurun_out/s3/never.err
TFREC_AMD_HOST_PROF=1 python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > gpurun_out/s3/hostprof.json 2> gpurun_out/s3/hostprof.err
