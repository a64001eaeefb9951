#!/bin/bash
# igemm3 x9 kernel: correctness against torch's own conv and per-shape time against the fp32-MFMA kernel
cd /root/repo
export PYTHONPATH=/root/repo
S="hg32 hg64 hg64q cv64_32 fe32 fe32q fe32d5 fe96_32 fe64 fe192_64 anm64d8 off81"
echo "== x9 on"; timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | tail -20
echo "== x9 off"; DPF_IGEMM3=0 timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | tail -20
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv and not deform" 2>&1 | tail -5
