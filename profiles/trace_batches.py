#!/usr/bin/env python3
"""usage: trace_batches.py kernel_trace.csv -- per batch of the timed region of a 20-step bench run (gate 1 + warm-up 5 + 20 front ends): start-end (ms since the
first timed front end) and hardware queue of each stage's kernel (profiles/r06_first_batch_timeline.txt, r06_host_stalls.txt)."""
import csv,re,collections,sys
def load(fn):
    rows=list(csv.DictReader(open(fn)))
    for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
    rows.sort(key=lambda r:r['s'])
    return rows
def short(n):
    n=re.sub(r'^void ','',n).replace('tfrec::','')
    n=re.sub(r'\(.*$','',n).replace('_kernel','')
    return n
rows=load(sys.argv[1])
fe=[r for r in rows if 'frontend_kernel' in r['Kernel_Name']]
# the timed region = the last run of 20 front ends before the final single (after gate) one; find via gaps: take fe[-21:-1]
timed=fe[6:26]
T0=timed[0]['s']
sel=[r for r in rows if r['s']>=T0-2000 and r['s']<=fe[26]['s']-1]
cnt=collections.Counter()
per=collections.defaultdict(dict)
for r in sel:
    k=short(r['Kernel_Name']); cnt[k]+=1
    per[cnt[k]][k]=((r['s']-T0)/1e6,(r['e']-T0)/1e6, r['Queue_Id'])
keys=['frontend<false>','fmdev','windows','spec_biquad<false, 0>','spec_biquad<true, 0>','whb_demod<false, false>','whb_chain','whb_check','mark','commit']
print('batch '+' | '.join('%-22s'%k[:22] for k in keys))
for b in range(1,21):
    print('%5d '%b+' | '.join(('%7.2f-%7.2f q%-3s'%per[b][k] if k in per[b] else ' '*22)+' ' for k in keys))
