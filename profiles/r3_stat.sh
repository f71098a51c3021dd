cd $GRAFT_REPO_ROOT
TFREC_AMD_LIB=$PWD/tfrec_amd/ab_stat.so python bench.py --steps 6 --warmup 1 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs 2>&1 | grep COOPSTAT
