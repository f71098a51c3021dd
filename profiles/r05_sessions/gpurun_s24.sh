#!/bin/bash
# session 24: front end with 8 outputs per thread (+ padded LDS image), discriminator pass with integer cross terms, one
# Newton step, scaled coefficients and ONE exact-direction test per lane: parity, counters, A/B against the previous build
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s24
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s24/pytest.txt
cd /tmp
for lib in default prev; do
	L=$R/tfrec_amd/libtfrec_amd.so; [ $lib = prev ] && L=$R/tfrec_amd/ab/prev.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s24/pmc_$lib.txt
done
cd $R
python profiles/ab_run.py gpurun_out/s24/ab.jsonl 3 100 8 new=default old=prev fe4=fe4 nopad=nopad > gpurun_out/s24/ab.txt 2>&1
for seed in 1101 1102; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s24/campaign.txt; done
