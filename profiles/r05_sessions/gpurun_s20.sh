#!/bin/bash
# session 20: config 5 with the 10:1 stage on a stream of its own
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s20
for m in 0 1 2 3; do
	TFREC_AMD_DECIM_OWN=$m timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5" 2>&1 | tail -1 >> gpurun_out/s20/pytest.txt
done
for r in 1 2; do for m in 0 1 2 3; do
	TFREC_AMD_DECIM_OWN=$m python bench.py --input-10x --streams 512 --steps 16 --warmup 4 --cpu-budget 0 --h2d-steps 0 --parity-streams 16 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:8]
print('DECIM_OWN=$m  %7.3f ms/step steady %s  parity %s  whole-path frac %.4f  %s' % (j['ms_per_step'], j.get('ms_per_step_steady'), j['config']['parity_ok'], j['roofline']['whole_path_frac'], ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
" >> gpurun_out/s20/config5.txt 2>&1
done; done
