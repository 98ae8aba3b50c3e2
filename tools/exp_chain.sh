cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
for e in "$@"; do
  rm -rf /tmp/xp; REVS_LIB=$R/tune/librevs_$e.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xp -o s -- python3 $R/tools/regime_run.py --regime ${REGIME:-binary} --steps 100 > /tmp/xp.log 2>&1
  echo "== $e: $(grep 'ms per iteration' /tmp/xp.log | cut -c1-50)"
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/xp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'agent_step_kernel' in r['Name'] and r['Name'].rstrip().endswith('true>(revs::AgentArgs)') and ', false, true>' in r['Name'] or 'op_chain_kv' in r['Name']:
        print('   ', r['Name'][:60], r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,2), 'min', round(float(r['MinNs'])/1e3,2), 'max', round(float(r['MaxNs'])/1e3,2))
PY
done
