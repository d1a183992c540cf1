cd /tmp && export TMPDIR=/tmp
for m in own rocprim; do
rm -rf /tmp/sp_$m; VGICP_PREP_SORT=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$m -o f -- python3 $GRAFT_REPO_ROOT/tools/probe_frame.py 20 60000 > /dev/null 2>&1
echo "== $m"; python3 - $m <<'PY'
import csv,glob,sys
for f in glob.glob(f"/tmp/sp_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(f"{float(r['AverageNs'])/1e3:9.2f} us x {r['Calls']:>6}  {r['Name'][:90]}")
PY
done
