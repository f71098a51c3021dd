#!/bin/bash
# session 5: full GPU suite on the round's build (256-slot segments); resource sensitivity: what do the long-lived kernels' LDS / register footprints cost the period?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s5/pytest.txt
python profiles/ab_run.py gpurun_out/s5/ab_sens.jsonl 2 60 8 \
  base=default \
  slicer_lds8k=default,TFREC_AMD_LDS_PAD_SLICER=8192 \
  slicer_lds24k=default,TFREC_AMD_LDS_PAD_SLICER=24576 \
  spec_lds8k=default,TFREC_AMD_LDS_PAD_SPEC=8192 \
  spec_lds24k=default,TFREC_AMD_LDS_PAD_SPEC=24576 \
  slicer_v175=slicer_v175 \
  spec_v250=spec_v250 \
  > gpurun_out/s5/ab_sens.txt 2>&1
