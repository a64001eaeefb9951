#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh grbm GRBM_GUI_ACTIVE -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh fetch FETCH_SIZE -- tools/conv_shape_bench.py hg32 fe32
bash tools/gpu_pmc.sh write WRITE_SIZE -- tools/conv_shape_bench.py hg32 fe32
