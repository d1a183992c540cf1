cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 30 60000 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/fr/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('vgicp::(anonymous namespace)::','').replace('void ','').split('(')[0][:60]
    if 'knn' in n or 'split' in n or 'cell_build' in n: print(n.ljust(62), r['Calls'].rjust(5), '%8.1f us'%(float(r['AverageNs'])/1e3),'min %.1f'%(float(r['MinNs'])/1e3))
PY
cd $GRAFT_REPO_ROOT
timeout 400 python3 tools/soak_preprocess.py 150 77 2>&1 | tail -2
