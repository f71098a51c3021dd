#!/usr/bin/env python3
"""usage: trace_window.py kernel_trace.csv <from ms> <to ms> -- every dispatch that starts inside the window (ms since the first timed front end of a 20-step bench run)."""
import csv,re,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
fe=[r for r in rows if 'frontend_kernel' in r['Kernel_Name']]
T0=fe[6]['s']
a,b=float(sys.argv[2]),float(sys.argv[3])
def short(n):
    n=re.sub(r'^void ','',n).replace('tfrec::','')
    return re.sub(r'\(.*$','',n).replace('_kernel','')
cnt=collections.Counter()
for r in rows:
    if r['s']<T0-2000: continue
    k=short(r['Kernel_Name']); cnt[k]+=1
    t=(r['s']-T0)/1e6
    if a<=t<=b: print("%7.2f-%7.2f %5.2f q%-2s %s #%d g=%s"%(t,(r['e']-T0)/1e6,(r['e']-r['s'])/1e6,r['Queue_Id'],k,cnt[k],r['Grid_Size_X']))
