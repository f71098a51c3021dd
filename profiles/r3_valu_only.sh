cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/run_valu.sh r3x > /dev/null 2>&1
grep -E "kernel  |slicer|decode" gpurun_out/r3x_pmc_valu.txt | cut -c1-330
