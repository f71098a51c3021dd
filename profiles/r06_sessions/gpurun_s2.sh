#!/bin/bash
# round 6 session 2: is stochastic PC sampling (rocprofv3 --pc-sampling-beta-enabled) usable on this pool?  It samples the
# waves of the OVERLAPPED pipeline (no dispatch serialisation) with an issue / stall reason per sample: VERDICT r05 item 2.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s2
cd /tmp
rm -rf /tmp/pcs
( timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 \
	--kernel-trace --output-format csv -d /tmp/pcs -- python $R/bench.py --steps 40 --warmup 6 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs \
	> $R/gpurun_out/s2/bench.json 2> $R/gpurun_out/s2/rocprof.err ) ; echo "rc=$?" > $R/gpurun_out/s2/rc.txt
find /tmp/pcs -type f | xargs ls -la >> $R/gpurun_out/s2/rc.txt 2>&1
for f in $(find /tmp/pcs -name '*pc_sampling*'); do
	head -5 $f > $R/gpurun_out/s2/head_$(basename $f).txt
	sz=$(stat -c %s $f)
	if [ $sz -lt 400000000 ]; then gzip -c $f > $R/gpurun_out/s2/$(basename $f).gz; fi
done
ls -la $R/gpurun_out/s2 >> $R/gpurun_out/s2/rc.txt
# keep what is merged back below 60 MB
find $R/gpurun_out/s2 -size +55M -delete
cp $(find /tmp/pcs -name '*kernel_trace.csv' | head -1) /tmp/kt.csv 2>/dev/null && gzip -c /tmp/kt.csv > $R/gpurun_out/s2/kernel_trace.csv.gz
find $R/gpurun_out/s2 -size +30M -name 'kernel_trace*' -delete
exit 0
