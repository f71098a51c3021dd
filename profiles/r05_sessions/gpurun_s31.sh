#!/bin/bash
# session 31: lane-per-step cooperative slicers with long emissions (TFA_1) and up to 16 rounds (TFA_2 family): VECSTAT, the new
# on/off test first, GPU suite, A/B against commit 43f02ce (head.so) 5 rounds, counters, campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s31
TFREC_AMD_LIB=$R/tfrec_amd/ab/vecstat.so python bench.py --steps 6 --warmup 2 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs 2>&1 | grep VECSTAT > gpurun_out/s31/vecstat.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "cooperative_slicers or bits_mode" 2>&1 | tail -15 > gpurun_out/s31/pytest_new.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s31/pytest.txt
python profiles/ab_run.py gpurun_out/s31/ab.jsonl 5 100 8 new=default old=head > gpurun_out/s31/ab.txt 2>&1
python - > gpurun_out/s31/stats.txt 2>&1 <<'P'
import json
for l in open("gpurun_out/s31/ab.jsonl"):
    j = json.loads(l)
    print(j["_label"], j["ms_per_step"], j["roofline"]["speculation_stats"], j["roofline"]["kernels_ms"].get("tfa1_coop_slicer_kernel"), j["roofline"]["kernels_ms"].get("coop_slicer_kernel"))
P
cd /tmp
for lib in default; do
	L=$R/tfrec_amd/libtfrec_amd.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s31/pmc_$lib.txt
	cp $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1) $R/gpurun_out/s31/pmc_$lib.csv
done
cd $R
for seed in 1205 1206 1207; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s31/campaign.txt; done
