# the single-pass query kernel (query_fused_kernel<true>, context option direct_query_result) against the default two-pass form:
# wall time per batch of tools/query_leg.py and the kernels' durations (rocprofv3); with the result block in device memory
# (direct_query_lds_kb = -1: timing only) and with fewer queries resident at once (LDS per workgroup in KB)
cd /tmp; export TMPDIR=/tmp
run() {
  rm -rf /tmp/qt; "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qt -o q -- python $GRAFT_REPO_ROOT/tools/query_leg.py 2>&1 | grep "query batches"
  python3 -c "
import csv
for r in csv.DictReader(open('/tmp/qt/q_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('query_fused_kernel', 'query_offsets', 'query_pack', 'copyBuffer')):
        print('   %-40s calls %3s  avg %8.1f us' % (r['Name'].replace('pgr::(anonymous namespace)::', '').split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
"
}
echo "== two passes (default)"; run env
for kb in 0 -1 9 18 36; do echo "== single pass, direct_query_lds_kb $kb"; run env PGR_DIRECT_QUERY_RESULT=1 PGR_DIRECT_QUERY_LDS_KB=$kb; done
