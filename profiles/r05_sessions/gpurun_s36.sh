#!/bin/bash
# session 36: whb_demod_kernel<false> inside the pipeline: when its 1024 one-wave workgroups start, how long the slowest stream takes
# (-DTFREC_AMD_PROFILE_WHB -DTFREC_AMD_PROFILE_WHB_SPAN), three runs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s36
for i in 1 2 3; do
	TFREC_AMD_LIB=$R/tfrec_amd/ab/whbspan.so python profiles/ubench/whb_cycles.py 2f pipelined span 2>&1 | grep -v amdgpu.ids >> gpurun_out/s36/whb_span.txt
done
TFREC_AMD_LIB=$R/tfrec_amd/ab/whbspan.so python profiles/ubench/whb_cycles.py 2f alone span 2>&1 | grep -v amdgpu.ids >> gpurun_out/s36/whb_span.txt
