#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="off81 off81a anm96d2 fe96_32 fe192_64 anm64d8"
echo "== x9"; timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | grep -v MIOpen | tail -20
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-400
