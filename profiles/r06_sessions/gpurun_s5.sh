#!/bin/bash
# round 6 session 5: fmdev_kernel decides from an fp32 estimate of the scaled angle where that is at least 1/64 from every integer,
# the rest (3 %) per workgroup through the fp64 path: fm tests, full GPU suite, A/B against the round's baseline, counters, campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s5
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fm_dev" 2>&1 | tail -15 > $O/pytest_fm.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 new=default old=base > $O/ab.txt 2>&1
cd /tmp
for lib in new; do
	L=$R/tfrec_amd/libtfrec_amd_exp.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $O/pmc_$lib.txt
done
cd $R
for seed in 6201 6202; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
exit 0
