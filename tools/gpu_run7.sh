#!/bin/bash
cd $GRAFT_REPO_ROOT
export PMC_FILTER="dcn_bwd_input"
bash tools/gpu_pmc.sh d1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- tools/dcn_bench.py all 64
bash tools/gpu_pmc.sh d2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -- tools/dcn_bench.py all 64
bash tools/gpu_pmc.sh d3 GRBM_GUI_ACTIVE -- tools/dcn_bench.py all 64
