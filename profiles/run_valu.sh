#!/bin/bash
# usage (GPU box, repo root): profiles/run_valu.sh <tag>  -> gpurun_out/<tag>_pmc_valu.txt, <tag>_pmc_mix.txt
# Counter-derived VALU / SALU busy per kernel (PMC passes only: no trace domains; rocprofv3 serialises the dispatches,
# so these are the kernels' own figures, not the co-scheduled pipeline's).  SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES /
# SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-valu}
cd /tmp
for pass in "valu SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
	set -- $pass
	name=$1; shift
	rm -rf /tmp/pmc_${tag}_$name
	rocprofv3 --pmc $* --output-format csv -d /tmp/pmc_${tag}_$name -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/${tag}_pmc_$name.log 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_${tag}_$name/*/*counter_collection.csv | head -1)) > $R/gpurun_out/${tag}_pmc_$name.txt
done
cut -c1-300 $R/gpurun_out/${tag}_pmc_valu.txt
