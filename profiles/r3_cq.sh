cd $GRAFT_REPO_ROOT
B="--steps 100 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
for v in 0 1 0 1 1; do TFREC_AMD_COPY_STREAM=$v python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('copystream $v', j['ms_per_step'], j['ms_per_step_steady'], j['config']['parity_ok'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.6))
"; done
