#!/bin/bash
# round 6 session 8: the three changes to the batch's serial stage loops together -- whb_zsr_kernel (the speculative average's zero-state
# response ahead of whb_demod_kernel<false>), the check as a chain per lane, the TFA_2 family's speculative biquad pass on its own stream
# (round 5's exact, zero-sum-by-itself change) -- and each of them switched off: WHB tests, GPU suite, A/B 100 steps, counters, campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s8
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady or bits" 2>&1 | tail -25 > $O/pytest_whb.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 2 100 8 all=default nozsr=default,TFREC_AMD_WHB_ZSR=0 nospec=default,TFREC_AMD_SPEC_OWN=0 rows=default,TFREC_AMD_WHB_CHECK_ROWS=1 old=base > $O/ab.txt 2>&1
python - > $O/stats.txt 2>&1 <<'P'
import json
for l in open("gpurun_out/s8/ab.jsonl"):
    j = json.loads(l)
    print(j["_label"], j["ms_per_step"], j["roofline"]["speculation_stats"].get("whb_respeculated"), j["roofline"]["kernels_ms"])
P
cd /tmp
for lib in new; do
	L=$R/tfrec_amd/libtfrec_amd_exp.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $O/pmc_$lib.txt
done
cd $R
for seed in 6401 6402; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
