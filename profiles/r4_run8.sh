#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench20b.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4_bench20b.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['ms_per_step_steady'], j['value'], j['roofline']['frac'], j['roofline']['kernel'], j['config']['parity_after_timed'])
for k,v in j['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('hbm_frac'), v.get('parity_ok'))
PY
