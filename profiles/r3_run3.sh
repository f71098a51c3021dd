cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r3c_pytest.txt
cat gpurun_out/r3c_pytest.txt
timeout 200 profiles/ubench/valu_issue > gpurun_out/r3c_valu_issue.jsonl 2> gpurun_out/r3c_valu_issue.err; echo "rc=$?" >> gpurun_out/r3c_valu_issue.err
B="--steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 64 --no-extra-configs"
for r in 1 2; do
python bench.py $B > gpurun_out/r3c_bench_spec$r.json 2>gpurun_out/r3c_bench_spec$r.err
TFREC_AMD_WHB_EXACT=1 python bench.py $B > gpurun_out/r3c_bench_exact$r.json 2>/dev/null
done
python bench.py $B --types 20 > gpurun_out/r3c_bench_whbonly.json 2>/dev/null
TFREC_AMD_VERIFY_LANES=16 python bench.py $B > gpurun_out/r3c_bench_v16.json 2>/dev/null
for f in spec1 exact1 spec2 exact2 whbonly v16; do python -c "
import json
j=json.loads(open('gpurun_out/r3c_bench_$f.json').read().strip().splitlines()[-1])
k=j['roofline']['kernels_ms']
print('$f', j['ms_per_step'], j['ms_median'], j['roofline']['speculation_stats'], ' '.join('%s=%.2f'%(a.replace('_kernel',''),b) for a,b in sorted(k.items(), key=lambda kv:-kv[1]) if b>0.05))
"; done
tail -3 gpurun_out/r3c_bench_spec1.err
cat gpurun_out/r3c_valu_issue.err; cat gpurun_out/r3c_valu_issue.jsonl
