#!/bin/bash
# round 6 session 6: the WHB check as an exact chain per LANE (whb_check.h: whb_chain_kernel + whb_check_kernel over the filter's
# input sequence) instead of a stream per row of 16 lanes: WHB tests, full GPU suite, A/B lanes / rows / round baseline, counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s6
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady or bits" 2>&1 | tail -25 > $O/pytest_whb.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 lanes=default rows=default,TFREC_AMD_WHB_CHECK_ROWS=1 old=base > $O/ab.txt 2>&1
cd /tmp
for lib in new; do
	L=$R/tfrec_amd/libtfrec_amd_exp.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $O/pmc_$lib.txt
done
cd $R
for seed in 6301 6302; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
