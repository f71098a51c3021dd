cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q -k "frontend or config5 or smoke or all_flush or auto_thresh or stress or hostile or wide" 2>&1 | tail -3
bash profiles/ab_driver.sh 2 > gpurun_out/r4_ab_fe.txt 2>&1; cat gpurun_out/r4_ab_fe.txt
