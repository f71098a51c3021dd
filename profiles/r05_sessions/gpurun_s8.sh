#!/bin/bash
# session 8: is the path bound by the memory system?  Extra coalesced HBM traffic beside every batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
for gb in 0 1 2 4 8 0 4; do
	python bench.py --bg-traffic-gb $gb --steps 60 --warmup 8 --cpu-budget 0 --h2d-steps 0 --parity-streams 8 --no-extra-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['roofline']['kernels_ms']
top = sorted(k.items(), key=lambda kv: -kv[1])[:8]
print('bg %4s GB per batch: %7.3f ms/step steady %s  %s' % ('$gb', j['ms_per_step'], j.get('ms_per_step_steady'), ' '.join('%s=%.2f' % (a.replace('_kernel',''), b) for a, b in top)))
" >> gpurun_out/s8/bg.txt 2>&1
done
# the copy alone: its own rate
python - >> gpurun_out/s8/bg.txt 2>&1 <<'PY'
import torch, time
a = torch.empty(2_000_000_000, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print('copy of 2 GB alone (4 GB of traffic): %.3f ms = %.2f TB/s' % (dt * 1e3, 4e9 / dt / 1e12))
PY
