#!/bin/bash
# session 25: discriminator pass back to one exact-direction test per sample (new arithmetic kept), buffer rotation of the biquad passes outside the divergent branch: parity, counters, A/B (prev = round-5 tree before these, nobq = without the biquad change, fe4 = 4 outputs per thread in the front end)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s25
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s25/pytest.txt
cd /tmp
for lib in default prev; do
	L=$R/tfrec_amd/libtfrec_amd.so; [ $lib = prev ] && L=$R/tfrec_amd/ab/prev.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s25/pmc_$lib.txt
done
cd $R
python profiles/ab_run.py gpurun_out/s25/ab.jsonl 3 100 8 new=default old=prev nobq=nobq fe4=fe4 > gpurun_out/s25/ab.txt 2>&1
for seed in 1103 1104; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s25/campaign.txt; done
