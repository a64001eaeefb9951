#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="hg32 hg64 cv64_32 fe32 fe96_32 off81 anm96d2 hg_s2"
echo "== g through registers"; timeout 600 python tools/conv_shape_bench.py --check $S 2>&1 | grep -v MIOpen | sed 's/fwd [0-9.]* ms.*| wgrad/| wgrad/' | tail -10
