#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "deform or full_size" 2>&1 | tail -3
echo "== new (offset rs)"; python tools/dcn_bench.py all 2>&1 | grep -v amdgpu
echo "== offset old"; DPF_DCN_OFF_RS=0 python tools/dcn_bench.py all 2>&1 | grep -v amdgpu
