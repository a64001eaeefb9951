#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "conv_forward_backward or conv_transpose3d" 2>&1 | tail -15 > gpurun_out/r2_t2.txt
echo "== default (check)" > gpurun_out/r2_b2.txt
python tools/conv_shape_bench.py --check >> gpurun_out/r2_b2.txt 2>&1
echo "== old wgrad" >> gpurun_out/r2_b2.txt
DPF_WGRAD2=0 python tools/conv_shape_bench.py hg32 hg64 fe32 fe64 off81 cv64_32 >> gpurun_out/r2_b2.txt 2>&1
echo "== WTH=2" >> gpurun_out/r2_b2.txt
DPF_W2_WTH=2 python tools/conv_shape_bench.py hg32 hg64 hg_s2 fe32 fe32q fe64 fe96_32 off81 cv64_32 >> gpurun_out/r2_b2.txt 2>&1
echo "== WTH=4 WNT=2 (forced, lds 160k)" >> gpurun_out/r2_b2.txt
DPF_W2_WTH=4 DPF_W2_WNT=2 python tools/conv_shape_bench.py hg32 hg64 fe32 fe32q fe64 fe96_32 off81 cv64_32 >> gpurun_out/r2_b2.txt 2>&1
echo "== WTH=2 WNT=1" >> gpurun_out/r2_b2.txt
DPF_W2_WTH=2 DPF_W2_WNT=1 python tools/conv_shape_bench.py hg32 hg64 hg_s2 fe32 fe64 off81 >> gpurun_out/r2_b2.txt 2>&1
echo "== blocks 1024" >> gpurun_out/r2_b2.txt
DPF_W2_BLOCKS=1024 python tools/conv_shape_bench.py hg32 hg64 fe32 fe64 off81 >> gpurun_out/r2_b2.txt 2>&1
echo "== blocks 512" >> gpurun_out/r2_b2.txt
DPF_W2_BLOCKS=512 python tools/conv_shape_bench.py hg32 hg64 fe32 fe64 off81 >> gpurun_out/r2_b2.txt 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/r2_t2.txt gpurun_out/r2_b2.txt
