cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "preprocess or prepare or knn or scan or frame or chain" 2>&1 | grep -E "passed|failed" | tail -2
bash tools/ab_prep.sh base tree
bash tools/ab_frame.sh base tree
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 30 60000 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/fr/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('vgicp::(anonymous namespace)::','').replace('void ','').split('(')[0][:60]
    if 'scan' in n or 'split' in n or 'cell_build' in n or 'knn' in n: print(n.ljust(62), r['Calls'].rjust(5), '%8.1f us'%(float(r['AverageNs'])/1e3),'min %.1f'%(float(r['MinNs'])/1e3))
PY
