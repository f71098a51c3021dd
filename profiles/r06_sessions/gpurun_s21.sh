#!/bin/bash
# round 6 session 21: the front end's hold on the workgroup dispatcher (profiles/r06_final_steps.txt: while its 196 k workgroups drain at high priority,
# no kernel of another stream STARTS).  (a) its stream confined to n compute units by a CU mask (normal priority); (b) n persistent workgroups that take
# the tiles in turn (TFREC_AMD_FE_PERSIST), at high and at normal stream priority; 60 steps each, parity on 8 streams
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/s21
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "frontend or steady" 2>&1 | tail -4 > $O/pytest_fe.txt
export AB_BENCH_ARGS="--cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs"
python profiles/ab_run.py $O/ab.jsonl 1 60 8 default=default \
	p1024=default,TFREC_AMD_FE_PERSIST=1024 p2048=default,TFREC_AMD_FE_PERSIST=2048 p4096=default,TFREC_AMD_FE_PERSIST=4096 p512=default,TFREC_AMD_FE_PERSIST=512 \
	p2048n=default,TFREC_AMD_FE_PERSIST=2048,TFREC_AMD_PRIO_FS=0 p1024n=default,TFREC_AMD_FE_PERSIST=1024,TFREC_AMD_PRIO_FS=0 \
	cu96=default,TFREC_AMD_FS_CUS=96 cu128=default,TFREC_AMD_FS_CUS=128 cu160=default,TFREC_AMD_FS_CUS=160 cu256=default,TFREC_AMD_FS_CUS=256 default2=default > $O/ab.txt 2>&1
exit 0
