#!/bin/bash
# session 1: GPU suite with the new parity columns, convergence histogram, segment-length A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s1/pytest.txt
TFREC_AMD_LIB=$PWD/tfrec_amd/ab/ck1seg1024.so TFREC_AMD_DEBUG_CONVHIST=1 TFREC_AMD_SPEC_DIV=1 TFREC_AMD_REPAIR_DIV=2 python bench.py --steps 6 --warmup 2 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > gpurun_out/s1/convhist.json 2> gpurun_out/s1/convhist.txt
python profiles/ab_run.py gpurun_out/s1/ab.jsonl 2 40 6 \
  base=default \
  seg256=seg256,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=6 \
  seg512=seg512,TFREC_AMD_SPEC_DIV=2,TFREC_AMD_REPAIR_DIV=3 \
  seg512r8=seg512,TFREC_AMD_SPEC_DIV=2,TFREC_AMD_REPAIR_DIV=8 \
  seg1024=seg1024,TFREC_AMD_SPEC_DIV=1,TFREC_AMD_REPAIR_DIV=2 \
  seg1024r8=seg1024,TFREC_AMD_SPEC_DIV=1,TFREC_AMD_REPAIR_DIV=8 \
  seg512s4=seg512,TFREC_AMD_SPEC_DIV=4,TFREC_AMD_REPAIR_DIV=8 \
  > gpurun_out/s1/ab.txt 2>&1
# the driver's command on the base build, for the fill/drain baseline
python bench.py --steps 20 --warmup 5 > gpurun_out/s1/driver_line.json 2> gpurun_out/s1/driver_line.err
