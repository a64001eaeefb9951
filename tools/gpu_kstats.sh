#!/bin/bash
# rocprofv3 per-kernel totals of any python command: tools/gpu_kstats.sh <tag> <script.py> [args]  -> top kernels by time
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/ks_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$@" > gpurun_out/ks_$tag.log 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv under $out (rocprofv3 failed, see the .log beside it)" >&2; exit 1; }
cp "$f" gpurun_out/ks_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%9.3f ms %6d calls %8.1f us avg  %5.1f%%  %s' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3,
                                                        100 * float(r['TotalDurationNs']) / tot, r['Name'][:110]))
print('total %.3f ms' % (tot / 1e6))
PY
