#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -k "deform or full_size or dcn or reference_fixture" 2>&1 | tail -3
echo "== fx cs8"; python tools/dcn_bench.py all 2>&1 | grep -v amdgpu
