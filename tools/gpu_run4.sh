#!/bin/bash
cd $GRAFT_REPO_ROOT
for cap in 256 384 512 768 1024; do
echo "== capacity $cap"
DPF_W2_CAPACITY=$cap python tools/conv_shape_bench.py hg32 hg64 off81 off81a anm96d2 fe32 fe64 2>&1 | grep -v "amdgpu.ids\|MIOpen" | sed 's/fwd.*dgrad[^|]*| //'
done
