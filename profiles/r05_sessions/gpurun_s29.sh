#!/bin/bash
# session 29: cooperative TFA_1 slicer with a step per lane (64 steps per pass): parity (GPU suite), fallback counts,
# counters and A/B against the tree of commit 43f02ce (head.so) and against its own scalar walk (TFREC_AMD_TFA1_VEC=0), campaign
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/s29
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s29/pytest.txt
python profiles/ab_run.py gpurun_out/s29/ab.jsonl 3 100 8 new=default old=head scalar=default,TFREC_AMD_TFA1_VEC=0 > gpurun_out/s29/ab.txt 2>&1
python - > gpurun_out/s29/stats.txt 2>&1 <<'P'
import json
for l in open("gpurun_out/s29/ab.jsonl"):
    j = json.loads(l)
    print(j["_label"], j["ms_per_step"], j["roofline"]["speculation_stats"], j["roofline"]["kernels_ms"].get("tfa1_coop_slicer_kernel"))
P
cd /tmp
for lib in default head; do
	L=$R/tfrec_amd/libtfrec_amd.so; [ $lib = head ] && L=$R/tfrec_amd/ab/head.so
	rm -rf /tmp/pmc_$lib
	TFREC_AMD_LIB=$L rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$lib -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1)) > $R/gpurun_out/s29/pmc_$lib.txt
	cp $(ls /tmp/pmc_$lib/*/*counter_collection.csv | head -1) $R/gpurun_out/s29/pmc_$lib.csv
done
cd $R
for seed in 1201 1202; do timeout 900 python tests/stress_gpu.py $seed 60 2>&1 | tail -1 >> gpurun_out/s29/campaign.txt; done
