#!/bin/bash
# round 6 session 3: which PC sampling configurations does the agent support (session 2: "not supported on any of the agents"
# for stochastic / cycles / 2^20)?  Then one attempt each: host_trap / time, stochastic / cycles at the listed interval.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3
mkdir -p $O
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3-avail list --pc-sampling > $O/avail_list.txt 2>&1
rocprofv3-avail info --pc-sampling > $O/avail_info.txt 2>&1
cd /tmp
try() {
	label=$1; shift
	rm -rf /tmp/pcs_$label
	( timeout 300 rocprofv3 --pc-sampling-beta-enabled "$@" --kernel-trace --output-format csv -d /tmp/pcs_$label -- \
		python $R/bench.py --steps 30 --warmup 6 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs \
		> $O/bench_$label.json 2> $O/err_$label.txt ); echo "$label rc=$?" >> $O/rc.txt
	find /tmp/pcs_$label -type f | xargs ls -la >> $O/rc.txt 2>&1
	for f in $(find /tmp/pcs_$label -name '*pc_sampling*'); do
		head -4 $f > $O/head_${label}_$(basename $f).txt
		sz=$(stat -c %s $f)
		if [ $sz -lt 300000000 ]; then gzip -c $f > $O/${label}_$(basename $f).gz; fi
	done
	f=$(find /tmp/pcs_$label -name '*kernel_trace.csv' | head -1)
	[ -n "$f" ] && gzip -c $f > $O/${label}_kernel_trace.csv.gz
}
try hosttrap --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100
try stoch16 --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536
find $O -size +28M -delete
du -sh $O >> $O/rc.txt
exit 0
