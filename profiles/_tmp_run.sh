cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print(j['ms_per_step'], j['ms_per_step_steady'], r['frac'], r['kernel'], r['chain_floor_ms'], r['chain_floor_kernel'], r['period_over_max_floor'], r['frontend'], r['context_memory'], r['traffic_ratio'], r['profiles'])
"
