#!/bin/bash
# round 6 session 41: which HIP call blocks inside submits 6 and 7 of the timed region: AMD_LOG_LEVEL=3 (API trace with microsecond stamps) of a short run,
# reduced on the box to the largest gaps between consecutive log lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/s41
mkdir -p $O
AMD_LOG_LEVEL=3 AMD_LOG_LEVEL_FILE=/tmp/hiplog python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --h2d-steps 0 --no-extra-configs --parity-streams 8 2>/dev/null | tail -1 > $O/line.json
ls -la /tmp/hiplog* > $O/ls.txt; gzip -c $(ls /tmp/hiplog* | head -1) > $O/hiplog.gz
python - > $O/gaps.txt <<'P'
import re,glob,json
j=json.loads(open('gpurun_out/s41/line.json').read()); print(j['ms_per_step'], j['host_ms']['submit_ms'])
fn=sorted(glob.glob('/tmp/hiplog*'))[0]
pat=re.compile(r'^:(\d):([^:]+):(\d+)\s*:\s*(\d+) us:\s*(.*)$')
prev=None; rows=[]
lines=open(fn,errors='replace').read().split('\n')
ts=[]
for i,l in enumerate(lines):
    m=pat.match(l)
    if m: ts.append((int(m.group(4)), i))
gaps=sorted(((ts[k+1][0]-ts[k][0], k) for k in range(len(ts)-1)), reverse=True)
print(len(lines), 'lines', len(ts), 'stamped')
# only gaps in the last part of the log (timed region ~ after the submit count passes warm-up): print the 30 largest gaps between 1 ms and 50 ms
n=0
for g,k in gaps:
    if g<1500 or g>60000: continue
    i0=ts[k][1]; i1=ts[k+1][1]
    print('---- gap %d us at line %d' % (g, i0))
    for l in lines[max(0,i0-3):i1+2]: print('   ', l[:230])
    n+=1
    if n>=24: break
P
head -c 3000 $O/gaps.txt
exit 0
