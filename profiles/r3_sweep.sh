cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 60 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 16 --no-extra-configs"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('%-22s'%'$name', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.3))
"
}
run base X=1
run cz TFREC_AMD_COOP_STREAM=1
run cz_q8 TFREC_AMD_COOP_STREAM=1 GPU_MAX_HW_QUEUES=8
run cznorm_q8 TFREC_AMD_COOP_STREAM=2 GPU_MAX_HW_QUEUES=8
run base_q8 GPU_MAX_HW_QUEUES=8
run cz_q16 TFREC_AMD_COOP_STREAM=1 GPU_MAX_HW_QUEUES=16
