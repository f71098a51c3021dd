#!/bin/bash
# round 6 session 4: whb_demod_kernel's candidate walk in scalar registers (spb = 64 as a constant, at most two candidates per
# step, sync search only while unsynced, entries stored directly): WHB tests, full GPU suite, A/B against the round's baseline
# (tfrec_amd/ab/base.so = commit 576301a), instruction counters of both, a few campaign rounds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s4
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or bits or steady" 2>&1 | tail -15 > $O/pytest_whb.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 new=default old=base > $O/ab.txt 2>&1
cd /tmp
for lib in new base; do
	if [ $lib = new ]; then L=$R/tfrec_amd/libtfrec_amd_exp.so; else L=$R/tfrec_amd/ab/base.so; fi
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $O/pmc_$lib.txt
done
cd $R
for seed in 6101 6102; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
