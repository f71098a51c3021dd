#!/bin/bash
# usage (GPU box, repo root): profiles/run_round.sh <tag>  -> gpurun_out/<tag>_* (copy what is to be kept into profiles/)
# bench line, kernel trace + stats of the same command, steady-state timeline, PMC passes (each in its own run, counters
# only: no trace domains), HBM traffic, VALU mix + measured issue costs -> the VALU roof.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04}
cd /tmp
python $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
python $R/bench.py --steps 200 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 64 > $R/gpurun_out/${tag}_bench_200steps.json 2>/dev/null
tail -1 $R/gpurun_out/${tag}_bench.json | cut -c1-400
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > $R/gpurun_out/${tag}_trace.log 2>&1
for f in /tmp/prof_$tag/*/*.db; do
	python $R/profiles/rocpd_summary.py $f > $R/gpurun_out/${tag}_kernel_stats.txt
	python $R/profiles/rocpd_timeline.py $f > $R/gpurun_out/${tag}_timeline.txt
	python $R/profiles/rocpd_steps.py $f 12 10 > $R/gpurun_out/${tag}_steps.txt   # two steady-state batches, with stream ids
done
cat $R/gpurun_out/${tag}_kernel_stats.txt
for pass in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "valu SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
	set -- $pass
	name=$1; shift
	rm -rf /tmp/pmc_${tag}_$name
	rocprofv3 --pmc $* --output-format csv -d /tmp/pmc_${tag}_$name -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/${tag}_pmc_$name.log 2>&1
done
for name in sq valu mix; do
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_${tag}_$name/*/*counter_collection.csv | head -1)) > $R/gpurun_out/${tag}_pmc_$name.txt
done
python $R/profiles/make_traffic.py /tmp/pmc_${tag}_fetch /tmp/pmc_${tag}_write $R/gpurun_out/${tag}_traffic.json 1024 48 47 > /dev/null
timeout 300 $R/profiles/ubench/valu_issue > $R/gpurun_out/${tag}_valu_issue.jsonl 2>/dev/null
python $R/profiles/make_valu.py $R/gpurun_out/${tag}_pmc_valu.txt $R/gpurun_out/${tag}_pmc_mix.txt $R/gpurun_out/${tag}_valu_issue.jsonl $R/gpurun_out/${tag}_valu.json
python -c "
import json; j=json.load(open('$R/gpurun_out/${tag}_traffic.json')); print('traffic total %.2f GB ratio %.2f' % (j['total_hbm_bytes_per_batch']/1e9, j['traffic_ratio']))"
rm -f $R/gpurun_out/${tag}_pmc_*.log
