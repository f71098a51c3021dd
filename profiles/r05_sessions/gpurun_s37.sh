#!/bin/bash
# session 37: what-if runs on the final tree (TFREC_AMD_SKIP leaves kernel groups out: results wrong, timing only; no parity gates):
# which stage loops the period sits on now
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s37
line() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$1', j['ms_per_step'], 'steady', j['ms_per_step_steady'], ' '.join('%s=%.2f'%(k.replace('_kernel',''),v) for k,v in sorted(j['roofline']['kernels_ms'].items(), key=lambda kv:-kv[1])[:9]))"; }
B="python bench.py --steps 60 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs"
for rep in 1 2; do
$B 2>/dev/null | line everything >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=16 $B 2>/dev/null | line no_whb_demod >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=32 $B 2>/dev/null | line no_whb_verify >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=48 $B 2>/dev/null | line no_whb_demod_no_verify >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=128 $B 2>/dev/null | line no_tfa2_biquads >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=176 $B 2>/dev/null | line no_tfa2_biquads_no_whb_demod_verify >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=10 $B 2>/dev/null | line no_tfa1_mark_slicers >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=5 $B 2>/dev/null | line no_tfa2_slicers >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=512 $B 2>/dev/null | line no_front_end >> gpurun_out/s37/whatif.txt
TFREC_AMD_SKIP=191 $B 2>/dev/null | line front_end_windows_whb_biquads_only >> gpurun_out/s37/whatif.txt
done
