#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <counters...> -- <python args>   (one --pmc pass; kernel-trace + stats only)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export TMPDIR=/tmp
tag=$1; shift
ctrs=()
while [ "${1:-}" != "--" ]; do [ $# -gt 0 ] || { echo "usage: $0 <tag> <counters...> -- <python args>" >&2; exit 2; }; ctrs+=("$1"); shift; done
shift
out=gpurun_out/pmc_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -o p -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find "$out" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter_collection.csv under $out (rocprofv3 failed, see the .log beside it)" >&2; exit 1; }
python3 tools/pmc_summary.py "$f" ${PMC_FILTER:-igemm2_kernel wgrad2_kernel conv_wgrad_kernel conv_igemm_kernel} > gpurun_out/pmc_$tag.txt
cat gpurun_out/pmc_$tag.txt
