#!/usr/bin/env python3
"""usage (in a directory holding kernel_trace<TAG>.csv of `rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 20 --warmup 5` and
that run's bench line bench<TAG>.json): trace_host_lag.py <TAG> -- per batch of the timed region: when the GPU finished the drain's copy, when the host's drain
returned (cumulated step_ms), the lag between them, and how long after the drain of batch b-4 the front end of batch b started (profiles/r06_host_stalls.txt)."""
import csv,re,json,sys
import numpy as np
tag=sys.argv[1]
rows=list(csv.DictReader(open('kernel_trace%s.csv'%tag)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
fe=[r for r in rows if 'frontend_kernel' in r['Kernel_Name']]
T0=fe[6]['s']; T1=fe[26]['s']
cp=[r for r in rows if 'copyBuffer' in r['Kernel_Name'] and r['Grid_Size_X']=='131072' and T0<r['s']<T1]
j=json.loads(open('bench%s.json'%tag).read().strip().splitlines()[-1])
st=np.cumsum(j['step_ms'])
fes=[(r['s']-T0)/1e6 for r in fe[6:26]]
print(tag, j['ms_per_step'], 'copies', len(cp))
print(' b  copy_end  host_done  lag   FE_start  (FE_start - host_done[b-4])')
for b in range(20):
    ce=(cp[b]['e']-T0)/1e6 if b<len(cp) else float('nan')
    print('%2d %8.2f %9.2f %6.2f %9.2f %s'%(b+1, ce, st[b], st[b]-ce, fes[b], ('%6.2f'%(fes[b]-st[b-4])) if b>=4 else ''))
