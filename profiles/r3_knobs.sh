cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 60 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
run() {
env TFREC_AMD_SHORT_TAILS=0 "$@" python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('$*', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.6))
"; }
run A=1
run TFREC_AMD_COOP_MIN=3072
run TFREC_AMD_SPEC_DIV=4
run TFREC_AMD_SPEC_DIV=16
run TFREC_AMD_FMDEV_K2=0
run TFREC_AMD_FMDEV_KW=1
run TFREC_AMD_REPAIR_DIV=6
run TFREC_AMD_REPAIR_DIV=24
run TFREC_AMD_SLICER_DIV=2
