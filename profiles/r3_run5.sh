cd $GRAFT_REPO_ROOT
for L in 16; do echo "noload lanes $L"; TFREC_AMD_VERIFY_LANES=$L TFREC_AMD_LIB=tfrec_amd/ab/pvnl.so python profiles/ubench/verify_cycles.py 20 2>&1 | tail -3; done
