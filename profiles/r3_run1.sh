cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r3a_pytest.txt
timeout 120 profiles/ubench/valu_issue > gpurun_out/r3a_valu_issue.jsonl 2> gpurun_out/r3a_valu_issue.err
python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > gpurun_out/r3a_bench_all.json 2>gpurun_out/r3a_bench_all.err
python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs --types 0f > gpurun_out/r3a_bench_nowhb.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs --types 20 > gpurun_out/r3a_bench_whb.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs --types 0e > gpurun_out/r3a_bench_tfa2.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs --types 21 > gpurun_out/r3a_bench_t1whb.json 2>/dev/null
timeout 600 profiles/run_valu.sh r3a > /dev/null 2>&1
cat gpurun_out/r3a_pytest.txt
for f in all nowhb whb tfa2 t1whb; do python -c "
import json,sys
j=json.loads(open('gpurun_out/r3a_bench_$f.json').read().strip().splitlines()[-1])
print('$f', j['ms_per_step'], j['ms_median'], j['value'])
"; done
cat gpurun_out/r3a_valu_issue.jsonl | head -50
