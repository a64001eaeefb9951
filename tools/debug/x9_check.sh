#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="hg32 cv64_32 fe32 fe32q fe96_32"
echo "== NT=2 for MT=1"; DPF_IGEMM3_NT=2 timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v MIOpen | tail -6
echo "== default"; timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v MIOpen | tail -6
