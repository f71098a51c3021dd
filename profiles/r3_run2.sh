cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r3b_pytest.txt
timeout 300 profiles/ubench/valu_issue > gpurun_out/r3b_valu_issue.jsonl 2> gpurun_out/r3b_valu_issue.err; echo "rc=$?" >> gpurun_out/r3b_valu_issue.err
cat gpurun_out/r3b_pytest.txt; cat gpurun_out/r3b_valu_issue.err; cat gpurun_out/r3b_valu_issue.jsonl
