cd $GRAFT_REPO_ROOT
timeout 600 profiles/run_steps.sh r3o > /dev/null 2>&1
tail -2 gpurun_out/r3o_steps.txt
