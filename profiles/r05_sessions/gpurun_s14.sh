#!/bin/bash
# session 14: the trimmed accepted-edge path of the cooperative TFA_2 slicer (+ the one-compare special test of fm_dev): parity, instruction counters, A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s14/pytest.txt
cd /tmp
for lib in default oldcoop; do
	L=$R/tfrec_amd/libtfrec_amd.so; [ $lib = oldcoop ] && L=$R/tfrec_amd/ab/oldcoop.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s14/pmc_$lib.txt
done
cd $R
python profiles/ab_run.py gpurun_out/s14/ab.jsonl 3 60 8 new=default old=oldcoop > gpurun_out/s14/ab.txt 2>&1
