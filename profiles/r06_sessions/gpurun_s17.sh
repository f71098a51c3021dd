#!/bin/bash
# round 6 session 17: whb_demod_kernel<false> with a HELPER wave per stream (the zero-state response of the next step through LDS, a barrier per
# step): WHB tests, GPU suite, A/B helper / roles swapped in every other workgroup / the commit before; counters; campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s17
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "whb or steady or bits" 2>&1 | tail -25 > $O/pytest_whb.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
python profiles/ab_run.py $O/ab.jsonl 3 100 8 helper=helper hswap=hswap nohelp=nohelp > $O/ab.txt 2>&1
cd /tmp
rm -rf /tmp/pmc_h
TFREC_AMD_LIB=$R/tfrec_amd/ab/helper.so timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_h -- python $R/bench.py --experiments --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_h/*/*counter_collection.csv | head -1)) > $O/pmc_helper.txt
cd $R
for seed in 6601 6602; do timeout 900 python tests/stress_gpu.py $seed 30 2>&1 | tail -1 >> $O/campaign.txt; done
exit 0
