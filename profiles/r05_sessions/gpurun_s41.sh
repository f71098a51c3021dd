#!/bin/bash
# session 41: whb_demod unrolled (unroll.so = the working tree) and, on top of it, the speculative biquad pass on its own stream (both.so),
# against the tree before (final.so): are the two zero-sum changes worth something together?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s41
TFREC_AMD_LIB=$R/tfrec_amd/ab/both.so timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady_state or config2_full_size or bits_mode or state_carries" 2>&1 | tail -5 > gpurun_out/s41/pytest_both.txt
python profiles/ab_run.py gpurun_out/s41/ab.jsonl 3 100 8 both=both unroll=unroll before=final > gpurun_out/s41/ab.txt 2>&1
