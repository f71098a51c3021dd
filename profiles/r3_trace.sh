export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace -d /tmp/prof_t -- python $R/bench.py --steps 30 --warmup 5 --cpu-budget 0 --h2d-steps 0 --parity-streams 0 --no-extra-configs > /dev/null 2>&1
for f in /tmp/prof_t/*/*.db; do python $R/profiles/rocpd_steps.py $f 12 9 > $R/gpurun_out/r3t_steps.txt; done
python - <<'PY'
import re,collections
rows=[]
for l in open('/root/repo/gpurun_out/r3t_steps.txt'):
    if l.startswith('#') or l.startswith('kernel'): continue
    m=re.match(r'(.+?)\s+(-?\d+\.\d+)\s+(-?\d+\.\d+)\s+(\d+\.\d+)\s+(.*)$',l.rstrip())
    if not m: continue
    rows.append((m.group(1).strip(),float(m.group(2)),float(m.group(3)),float(m.group(4)),m.group(5)))
by=collections.defaultdict(list)
for r in rows: by[r[4]].append(r)
for q,rs in sorted(by.items()):
    print('== stream/queue',q,' busy %.2f ms'%sum(r[3] for r in rs))
    print('   '+' | '.join('%s %.2f-%.2f'%(r[0][:16],r[1],r[2]) for r in rs if r[3]>0.05))
print(open('/root/repo/gpurun_out/r3t_steps.txt').read().splitlines()[-1])
PY
