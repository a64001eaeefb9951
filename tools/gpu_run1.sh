#!/bin/bash
# round-2 iteration script (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "conv_forward_backward or conv_transpose3d" 2>&1 | tail -15 > gpurun_out/r2_t1.txt
echo "== old kernel" > gpurun_out/r2_b1.txt
DPF_IGEMM2=0 python tools/conv_shape_bench.py hg32 hg_s2 hg64 fe32 fe64 off81 >> gpurun_out/r2_b1.txt 2>&1
echo "== igemm2 default (check)" >> gpurun_out/r2_b1.txt
python tools/conv_shape_bench.py --check >> gpurun_out/r2_b1.txt 2>&1
echo "== igemm2 CC=4" >> gpurun_out/r2_b1.txt
DPF_G2_CC=4 python tools/conv_shape_bench.py >> gpurun_out/r2_b1.txt 2>&1
echo "== igemm2 CC=2" >> gpurun_out/r2_b1.txt
DPF_G2_CC=2 python tools/conv_shape_bench.py >> gpurun_out/r2_b1.txt 2>&1
echo "== igemm2 CC=8" >> gpurun_out/r2_b1.txt
DPF_G2_CC=8 python tools/conv_shape_bench.py hg64 fe32 fe32q fe64 fe96_32 anm96d2 >> gpurun_out/r2_b1.txt 2>&1
echo "== igemm2 NT=4 (MT<=2)" >> gpurun_out/r2_b1.txt
DPF_G2_NT=4 python tools/conv_shape_bench.py hg_s2 hg64 hg64_s2 fe64 fe192_64 anm64d8 >> gpurun_out/r2_b1.txt 2>&1
cat gpurun_out/r2_t1.txt gpurun_out/r2_b1.txt
