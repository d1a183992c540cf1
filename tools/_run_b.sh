cd $GRAFT_REPO_ROOT
bash tools/ab_prep.sh tree h_b2 h_b8 h_b16 h_l5b16 h_l3b2 2>&1 | grep -v "^$"
cd /tmp && export TMPDIR=/tmp
for v in tree h_b2 h_b8 h_b16 h_l5b16 h_l3b2; do
  if [ "$v" = tree ]; then unset VGICP_LIB_PATH; else export VGICP_LIB_PATH=$GRAFT_REPO_ROOT/eskf_lio_amd/lib_ab/$v/libvgicp_hip.so; fi
  rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fr -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 20 60000 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/fr/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'knn_search' in r['Name']: print(sys.argv[1],'frame knn %.1f us (min %.1f)'%(float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3))
PY
done
