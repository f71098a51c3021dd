#!/bin/bash
# usage (GPU box, repo root): profiles/run_traffic.sh <tag>  -> gpurun_out/<tag>_traffic.json  (the two HBM counter passes only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-t}
cd /tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
	set -- $pass
	name=$1; shift
	rm -rf /tmp/pmc_${tag}_$name
	rocprofv3 --pmc $* --output-format csv -d /tmp/pmc_${tag}_$name -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > $R/gpurun_out/${tag}_pmc_$name.log 2>&1
done
python $R/profiles/make_traffic.py /tmp/pmc_${tag}_fetch /tmp/pmc_${tag}_write $R/gpurun_out/${tag}_traffic.json 1024 48 47
