#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0; do
rm -rf gpurun_out/prof_dcn$v
DPF_DCN_OFF_RS=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dcn$v -o p -- python3 tools/dcn_bench.py all > /dev/null 2>&1
echo "== OFF_RS=$v"
python3 - gpurun_out/prof_dcn$v/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dcn' in r['Name']:
        print('%-60s calls %3s avg %8.3f ms' % (r['Name'].replace('(anonymous namespace)::', '').split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
