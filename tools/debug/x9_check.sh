#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="hg32 hg64 cv64_32 fe32 fe32q fe96_32 fe64 off81 anm96d2 hg_s2 anm64d8"
echo "== vec store"; timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | grep -v MIOpen | tail -12
echo "== scalar store"; DPF_G2_VEC_STORE=0 timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v MIOpen | tail -12
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv and not deform" 2>&1 | tail -3
bash tools/debug/x9_stamps.sh
