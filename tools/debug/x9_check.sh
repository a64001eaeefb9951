#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
S="hg32 fe32 fe32q fe96_32 hg64"
echo "== default"; timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v -e MIOpen -e amdgpu.ids | tail -5
echo "== nt stores"; DPF_G2_VEC_STORE=2 timeout 600 python tools/conv_shape_bench.py $S 2>&1 | grep -v -e MIOpen -e amdgpu.ids | tail -5
for v in 1 2; do DPF_G2_VEC_STORE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-140; done
