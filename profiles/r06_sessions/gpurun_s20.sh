#!/bin/bash
# round 6 session 20: waits -- whb_chain_kernel's barriers wait for the waves' LDS operations only (not for their global loads and stores), whb_demod_kernel
# stores the filter's input sequence at the step's start (bar) against the commit before (head4): WHB tests, GPU suite, A/B 100 steps x 3, counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s20
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady or bits" 2>&1 | tail -5 > $O/pytest_whb.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 bar=bar head4=head4 > $O/ab.txt 2>&1
cd /tmp
rm -rf /tmp/pmc_b
TFREC_AMD_LIB=$R/tfrec_amd/ab/bar.so timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_b -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_b/*/*counter_collection.csv | head -1)) > $O/pmc_bar.txt
cd $R
for seed in 6901 6902; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
