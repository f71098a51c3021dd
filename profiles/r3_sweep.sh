cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 16 --no-extra-configs"
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('%-22s'%'$name', j['ms_per_step'], j['ms_median'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.3))
"
}
run base X=1
run coop2048 TFREC_AMD_COOP_MIN=2048
run coop1024 TFREC_AMD_COOP_MIN=1024
run coop8192 TFREC_AMD_COOP_MIN=8192
run head32 TFREC_AMD_HEAD_CHUNKS=32
run head16 TFREC_AMD_HEAD_CHUNKS=16
run head128 TFREC_AMD_HEAD_CHUNKS=128
run specdiv4 TFREC_AMD_SPEC_DIV=4
run specdiv16 TFREC_AMD_SPEC_DIV=16
run repdiv6 TFREC_AMD_REPAIR_DIV=6
run repdiv24 TFREC_AMD_REPAIR_DIV=24
run t1early TFREC_AMD_T1_EARLY=1
run base2 X=1
