#!/bin/bash
# session 4: length-class queues for the lane-per-window slicers; full suite on the round's default build; slicer knobs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s4/pytest.txt
python profiles/ab_run.py gpurun_out/s4/ab_class.jsonl 3 60 8 \
  class=default noclass=noclass class2=class2 \
  class_div2=default,TFREC_AMD_SLICER_DIV=2 \
  class_coop3072=default,TFREC_AMD_COOP_MIN=3072 \
  class_head32=default,TFREC_AMD_HEAD_CHUNKS=32 \
  > gpurun_out/s4/ab_class.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/s4/driver_line.json 2> gpurun_out/s4/driver_line.err
