cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="--steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
for r in 1 2; do
python bench.py $B > gpurun_out/r3l_bench_kw$r.json 2>/dev/null
TFREC_AMD_FMDEV_KW=0 python bench.py $B > gpurun_out/r3l_bench_k2$r.json 2>/dev/null
done
for f in kw1 k21 kw2 k22; do python -c "
import json
j=json.loads(open('gpurun_out/r3l_bench_$f.json').read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('$f', j['ms_per_step'], j['ms_median'], j['config']['parity_ok'], j['roofline']['speculation_stats']['whb_respeculated'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.05))
"; done
