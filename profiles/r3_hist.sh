cd $GRAFT_REPO_ROOT
TFREC_AMD_DEBUG_WINHIST=1 python bench.py --steps 6 --warmup 1 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs 2>&1 | grep WINHIST
