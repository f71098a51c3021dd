export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for v in w1 w2; do
rm -rf /tmp/pmc_cs
TFREC_AMD_LIB=$R/tfrec_amd/ab_$v.so rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_cs -- python $R/bench.py --steps 1 --warmup 1 --cpu-budget 0 --parity-streams 0 --h2d-steps 0 --no-extra-configs > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pmc_cs/*/*counter_collection.csv')[0]
d=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    if 'coop_slicer' in k:
        d[(int(r['Dispatch_Id']),k)][r['Counter_Name']]=float(r['Counter_Value'])
for (i,k),v in sorted(d.items())[:2]:
    print('$v',i,k[:40],' '.join('%s=%.4g'%(a,b) for a,b in sorted(v.items())))
PY
done
