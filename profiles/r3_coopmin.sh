cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 100 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
for t in 4096 3072 2048 1024 400; do TFREC_AMD_COOP_MIN=$t python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('coopmin $t', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.3))
"; done
