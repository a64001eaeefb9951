#!/bin/bash
# HBM traffic of the deformable-conv kernels alone (tools/dcn_bench.py): separate FETCH_SIZE / WRITE_SIZE passes -> per-launch bytes.
# usage (through gpurun): bash tools/dcn_traffic.sh <tag>
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export TMPDIR=/tmp
tag=${1:?tag}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "gpurun_out/dt_${tag}_$c"
  rocprofv3 --kernel-trace --pmc "$c" --output-format csv -d "gpurun_out/dt_${tag}_$c" -o p -- python3 tools/dcn_bench.py > /dev/null 2>&1
done
python3 tools/pmc_traffic.py "gpurun_out/dt_${tag}_FETCH_SIZE/p_counter_collection.csv" "gpurun_out/dt_${tag}_WRITE_SIZE/p_counter_collection.csv" "gpurun_out/dt_${tag}.json" 1 > /dev/null
python3 - "gpurun_out/dt_${tag}.json" <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    if k.startswith('dcn'):
        print('%-16s launches %3d  fetch x2 %6.2f GB  write %5.2f GB  total %6.2f GB per launch' % (
            k, v['launches'], v['fetch_bytes_per_launch_x2'] / 1e9, v['write_bytes_per_launch'] / 1e9, v['hbm_bytes_per_launch'] / 1e9))
PY
