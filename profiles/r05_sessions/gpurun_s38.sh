#!/bin/bash
# session 38: the speculative biquad pass off k2 (every segment from zero, the first one repaired from the carried state; PipeCtl::ks):
# biquad / steady-state / full-size tests first, then the GPU suite, A/B own stream (default) / on k2 (TFREC_AMD_SPEC_OWN=0) / the tree
# before (final.so), campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s38
timeout 900 python -m pytest tests -m gpu -x -q -k "state_carries or steady_state or config2_full_size or quarter_of_config2 or deep_and_shallow or bits_mode" 2>&1 | tail -5 > gpurun_out/s38/pytest_some.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/s38/pytest.txt
python profiles/ab_run.py gpurun_out/s38/ab.jsonl 3 100 8 own=default onk2=default,TFREC_AMD_SPEC_OWN=0 before=final > gpurun_out/s38/ab.txt 2>&1
python - > gpurun_out/s38/stats.txt 2>&1 <<'P'
import json
for l in open("gpurun_out/s38/ab.jsonl"):
    j = json.loads(l)
    print(j["_label"], j["ms_per_step"], j["roofline"]["speculation_stats"])
P
for seed in 1401 1402 1403; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s38/campaign.txt; done
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs > gpurun_out/s38/driver_line_$i.json 2>/dev/null; done
