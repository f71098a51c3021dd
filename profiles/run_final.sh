#!/bin/bash
# usage (GPU box, repo root): profiles/run_final.sh <tag>  -> gpurun_out/<tag>_* (copy what is to be kept into profiles/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-final}
cd /tmp
python $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
tail -1 $R/gpurun_out/${tag}_bench.json | cut -c1-600
# kernel trace + stats of the same command
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs > $R/gpurun_out/${tag}_trace.log 2>&1
for f in /tmp/prof_$tag/*/*.db; do
	python $R/profiles/rocpd_summary.py $f > $R/gpurun_out/${tag}_kernel_stats.txt
	python $R/profiles/rocpd_timeline.py $f > $R/gpurun_out/${tag}_timeline.txt
	python $R/profiles/rocpd_steps.py $f 12 10 > $R/gpurun_out/${tag}_steps.txt   # two steady-state batches, with stream ids
done
cat $R/gpurun_out/${tag}_kernel_stats.txt
# PMC passes (counters only: no trace domains)
for pass in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
	set -- $pass
	name=$1; shift
	rm -rf /tmp/pmc_${tag}_$name
	rocprofv3 --pmc $* --output-format csv -d /tmp/pmc_${tag}_$name -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/${tag}_pmc_$name.log 2>&1
done
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_${tag}_sq/*/*counter_collection.csv | head -1)) > $R/gpurun_out/${tag}_pmc_sq.txt
python $R/profiles/make_traffic.py /tmp/pmc_${tag}_fetch /tmp/pmc_${tag}_write $R/gpurun_out/${tag}_traffic.json 1024 48 47 > /dev/null
cat $R/gpurun_out/${tag}_pmc_sq.txt | cut -c1-250
