#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="fe32d5 anm64d8 anm96d2"
echo "== rstep"; timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | grep -v -e MIOpen -e amdgpu.ids | tail -4
echo "== rstep off"; DPF_IGEMM3_RSTEP=0 timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v -e MIOpen -e amdgpu.ids | tail -4
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv and not deform" 2>&1 | tail -3
