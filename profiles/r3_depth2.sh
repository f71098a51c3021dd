cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 80 --warmup 10 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
run() {
d=$1; shift
env "$@" python bench.py $B --depth $d 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('depth $d $*', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.6))
"; }
for d in 4 5 6; do run $d TFREC_AMD_REPAIR_KW=0; run $d TFREC_AMD_REPAIR_KW=1; done
