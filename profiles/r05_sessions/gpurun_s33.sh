#!/bin/bash
# session 33: default long window 1024; wave priorities of the two WHB kernels again now that the cooperative slicers are cheap
# (whb_demod_kernel<false> is the longest kernel inside the batch): default (1 / none), w3 (3 / none), w3v3 (3 / 3), w2v2 (2 / 2)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s33
timeout 600 python -m pytest tests -m gpu -x -q -k "cooperative_slicers or bits_mode or config2_full_size or steady_state" 2>&1 | tail -5 > gpurun_out/s33/pytest_some.txt
python profiles/ab_run.py gpurun_out/s33/ab.jsonl 2 100 8 new=default w3=w3 w3v3=w3v3 w2v2=w2v2 c4096=default,TFREC_AMD_COOP_MIN=4096 > gpurun_out/s33/ab.txt 2>&1
