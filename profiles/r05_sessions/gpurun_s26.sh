#!/bin/bash
# session 26: instruction-cache microbenchmark (profiles/ubench/icache.hip); mark_kernel whole-chunk path + mark_lvl / 2 as a
# shift: parity, counters, A/B against the tree of commit b25ab95 (prev2)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s26
timeout 300 profiles/ubench/icache > gpurun_out/s26/icache.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s26/pytest.txt
cd /tmp
for lib in default prev2; do
	L=$R/tfrec_amd/libtfrec_amd.so; [ $lib = prev2 ] && L=$R/tfrec_amd/ab/prev2.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQC_ICACHE_REQ SQC_ICACHE_MISSES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s26/pmc_$lib.txt
done
cd $R
python profiles/ab_run.py gpurun_out/s26/ab.jsonl 3 100 8 new=default old=prev2 > gpurun_out/s26/ab.txt 2>&1
for seed in 1105 1106; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s26/campaign.txt; done
