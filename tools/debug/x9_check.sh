#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform or transpose" 2>&1 | tail -3
echo "== dcn bench"; timeout 600 python tools/dcn_bench.py all 2>&1 | grep -v amdgpu.ids | tail -12
echo "== hg_s2"; timeout 300 python tools/conv_shape_bench.py hg_s2 hg64_s2 2>&1 | grep -v -e MIOpen -e amdgpu.ids | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-140; done
