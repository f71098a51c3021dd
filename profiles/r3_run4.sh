cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r3k_pytest.txt
cat gpurun_out/r3k_pytest.txt
B="--steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
python bench.py $B > gpurun_out/r3k_bench_spec1.json 2>gpurun_out/r3k_bench_spec1.err
TFREC_AMD_WHB_EXACT=1 python bench.py $B > gpurun_out/r3k_bench_exact1.json 2>/dev/null
TFREC_AMD_WHB_FORCE_FAIL=100 python bench.py $B > gpurun_out/r3k_bench_ff100.json 2>/dev/null
for f in spec1 exact1 ff100; do python -c "
import json
j=json.loads(open('gpurun_out/r3k_bench_$f.json').read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('$f', j['ms_per_step'], j['ms_median'], j['config']['parity_ok'], j['roofline']['speculation_stats']['whb_respeculated'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.05))
"; done
tail -3 gpurun_out/r3k_bench_spec1.err
