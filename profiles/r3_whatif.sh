cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 60 --warmup 15 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs"
for sk in 0 64 512 576 192 15 527; do
TFREC_AMD_SKIP=$sk python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('skip $sk', j['ms_per_step'], j['ms_per_step_steady'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.3))
"; done
