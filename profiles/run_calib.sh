#!/bin/bash
# usage (GPU box, repo root): profiles/run_calib.sh  -> gpurun_out/hbm_calibration.json (copy into profiles/r0N_hbm_calibration.json)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf /tmp/calib_$c
	rocprofv3 --pmc $c --output-format csv -d /tmp/calib_$c -- $R/profiles/ubench/hbm_calib > $R/gpurun_out/calib_$c.log 2>&1
done
python $R/profiles/make_calibration.py /tmp/calib_FETCH_SIZE /tmp/calib_WRITE_SIZE $R/gpurun_out/hbm_calibration.json
