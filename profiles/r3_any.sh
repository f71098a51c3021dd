cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash profiles/run_valu.sh r3y > /dev/null 2>&1
python - <<'PY'
import re
rows=[]; tot=0
for l in open('gpurun_out/r3y_pmc_valu.txt'):
    m=re.match(r'(\S.*?)\s+dispatches=(\d+)\s+(.*)',l)
    if not m or m.group(1).startswith('__'): continue
    d={k:float(v) for k,v in (kv.split('=') for kv in m.group(3).split())}
    per=int(m.group(2))/3
    rows.append((d['SQ_ACTIVE_INST_ANY']*per, d['SQ_INSTS_VALU']*per, d['SQ_INSTS_SALU']*per, d['GRBM_GUI_ACTIVE']/8/2.4e6*per, m.group(1)))
    tot+=d['SQ_ACTIVE_INST_ANY']*per
print('issue roof %.2f ms'%(tot*4/1024/2.4e6))
for a,v,s_,g,n in sorted(rows,reverse=True)[:16]: print('%.3g quad = %.2f ms | VALU %.3g SALU %.3g | alone %.2f ms | %s'%(a,a*4/1024/2.4e6,v,s_,g,n.replace('void ','').replace('tfrec::','')))
PY
