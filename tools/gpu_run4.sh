#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "conv_forward_backward or conv_transpose3d" 2>&1 | tail -15 > gpurun_out/r2_t4.txt
echo "== default (check)" > gpurun_out/r2_b4.txt
python tools/conv_shape_bench.py --check >> gpurun_out/r2_b4.txt 2>&1
echo "== blocks 768" >> gpurun_out/r2_b4.txt
DPF_W2_BLOCKS=768 python tools/conv_shape_bench.py hg32 hg64 hg_s2 fe32 fe32q fe64 fe96_32 off81 cv64_32 >> gpurun_out/r2_b4.txt 2>&1
echo "== NCT max 5" >> gpurun_out/r2_b4.txt
DPF_W2_NCT=5 python tools/conv_shape_bench.py hg32 hg64 hg_s2 fe32 fe32q fe64 fe96_32 off81 cv64_32 >> gpurun_out/r2_b4.txt 2>&1
echo "== NCT max 4 blocks 768" >> gpurun_out/r2_b4.txt
DPF_W2_NCT=4 DPF_W2_BLOCKS=768 python tools/conv_shape_bench.py hg32 hg64 hg_s2 fe32 fe32q fe64 fe96_32 off81 cv64_32 >> gpurun_out/r2_b4.txt 2>&1
grep -v "amdgpu.ids\|MIOpen" gpurun_out/r2_t4.txt gpurun_out/r2_b4.txt
bash tools/gpu_pmc.sh sq1b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh grbmb GRBM_GUI_ACTIVE -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh fetchb FETCH_SIZE -- tools/conv_shape_bench.py hg32 fe32
