cd $GRAFT_REPO_ROOT
B="--steps 100 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
run() {
env "$@" python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('$*', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.6))
"; }
run A=1
run TFREC_AMD_SLICER_DIV=2
run TFREC_AMD_SPEC_DIV=6
run TFREC_AMD_SPEC_DIV=12
run TFREC_AMD_REPAIR_DIV=8
run TFREC_AMD_REPAIR_DIV=16
run TFREC_AMD_HEAD_CHUNKS=32
run TFREC_AMD_COOP_BLOCKS=16384
