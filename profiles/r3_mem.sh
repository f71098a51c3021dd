export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for pass in "tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum GRBM_GUI_ACTIVE" \
            "ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "ic SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
	set -- $pass
	name=$1; shift
	rm -rf /tmp/pmc_m_$name
	rocprofv3 --pmc $* --output-format csv -d /tmp/pmc_m_$name -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/r3m_pmc_$name.log 2>&1
	python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_m_$name/*/*counter_collection.csv | head -1)) > $R/gpurun_out/r3m_pmc_$name.txt
	echo "== $name"; grep -v "^__" $R/gpurun_out/r3m_pmc_$name.txt | cut -c1-260
done
