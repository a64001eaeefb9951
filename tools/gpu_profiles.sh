#!/bin/bash
# Collects a round's evidence under gpurun_out/<tag>/ (copied to profiles/ afterwards): bench lines, per-shape table, rocprofv3 kernel
# stats, PMC traffic (separate FETCH / WRITE passes) and SQ counters of the dominant kernels.   usage: tools/gpu_profiles.sh r03 [quick]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export TMPDIR=/tmp
R=${1:?round tag, e.g. r03}
QUICK=${2:-}
O=gpurun_out/$R
mkdir -p "$O"
run() { "$@" 2>> "$O/bench.err"; }
# PMC traffic first (the bench line reads profiles/<R>_pmc_traffic.json when it exists)
pmc_pass() {   # <dir> <counter> <bench args...>
  local d=$1 c=$2; shift 2
  rm -rf "gpurun_out/$d"
  rocprofv3 --kernel-trace --pmc "$c" --output-format csv -d "gpurun_out/$d" -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-detail --wgrad-inline --no-graph "$@" > /dev/null 2>&1
}
pmc_pass pmc_f FETCH_SIZE; pmc_pass pmc_w WRITE_SIZE
python3 tools/pmc_traffic.py gpurun_out/pmc_f/p_counter_collection.csv gpurun_out/pmc_w/p_counter_collection.csv "$O/${R}_pmc_traffic.json" 3
gzip -c gpurun_out/pmc_f/p_counter_collection.csv > "$O/${R}_pmc_fetch_raw.csv.gz"; gzip -c gpurun_out/pmc_w/p_counter_collection.csv > "$O/${R}_pmc_write_raw.csv.gz"
mkdir -p profiles; cp "$O/${R}_pmc_traffic.json" profiles/
run python bench.py --steps 10 --warmup 3 --shapes "$O/${R}_conv_shape_table.txt" > "$O/${R}_bench.json"
bash tools/gpu_prof.sh "$R" > "$O/${R}_bench_family_ms.txt" 2>&1
cp "gpurun_out/prof_${R}_kernel_stats.csv" "$O/${R}_bench_kernel_stats.csv"
export PMC_FILTER="igemm3 igemm2 wgrad2 dcn_ pointwise bn_"
bash tools/gpu_pmc.sh ${R}sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-detail --wgrad-inline --no-graph > "$O/${R}_sq_counters.txt" 2>&1
if [ -z "$QUICK" ]; then
  pmc_pass pmc_f3 FETCH_SIZE --precision bf16; pmc_pass pmc_w3 WRITE_SIZE --precision bf16
  python3 tools/pmc_traffic.py gpurun_out/pmc_f3/p_counter_collection.csv gpurun_out/pmc_w3/p_counter_collection.csv "$O/${R}_pmc_traffic_bf16.json" 3
  cp "$O/${R}_pmc_traffic_bf16.json" profiles/
  run python bench.py --height 512 --width 768 --no-cpu-baseline > "$O/${R}_bench_c2_512x768.json"
  run python bench.py --model psmnet --batch 2 --no-cpu-baseline > "$O/${R}_bench_c4_psmnet_train.json"
  run python bench.py --precision bf16 --no-cpu-baseline --shapes "$O/${R}_conv_shape_table_bf16.txt" > "$O/${R}_bench_c5_bf16_b4.json"
  run python bench.py --precision bf16 --batch 8 --no-cpu-baseline > "$O/${R}_bench_c5_bf16_b8.json"
  run python bench.py --wgrad-inline --no-cpu-baseline > "$O/${R}_bench_wgrad_inline.json"
  run python bench.py --model nnet --batch 2 --no-cpu-baseline > "$O/${R}_bench_nnet_train.json"
  run python bench.py --model stereonet --no-cpu-baseline > "$O/${R}_bench_stereonet_train.json"
  run python bench.py --workload cost_volume --no-cpu-baseline > "$O/${R}_bench_cost_volume_stage.json"
  run python bench.py --workload cost_volume_fix --no-cpu-baseline > "$O/${R}_bench_cost_volume_fix_stage.json"
  run python bench.py --workload psm_volume --batch 2 --no-cpu-baseline > "$O/${R}_bench_psm_volume.json"
  bash tools/gpu_prof.sh ${R}bf16 --precision bf16 > "$O/${R}_bench_c5_bf16_family_ms.txt" 2>&1
  cp "gpurun_out/prof_${R}bf16_kernel_stats.csv" "$O/${R}_bench_c5_bf16_kernel_stats.csv"
  bash tools/gpu_kstats.sh ${R}dcn tools/dcn_bench.py all > "$O/${R}_dcn_bench_kernel_stats.txt" 2>&1
  # per-shape conv times: x9 kernels (default) against the exact-f32 matrix instruction, 16-byte against 4-byte tile stores
  SH9="hg32 hg64 hg64q cv64_32 fe32 fe32q fe32d5 fe96_32 fe64 fe192_64 anm96d2 anm64d8 off81 off81a hg_s2"
  { echo "== default (fp32 products from f16 components, 16-byte tile stores)"; python tools/conv_shape_bench.py --check $SH9 2>&1 | grep -v -e MIOpen -e amdgpu.ids
    echo "== DPF_F32_X9=1 (six bf16 partial products of exact three-way splits: the default before the f16 components)"; DPF_F32_X9=1 python tools/conv_shape_bench.py $SH9 2>&1 | grep -v -e MIOpen -e amdgpu.ids
    echo "== DPF_F32_X9=0 (v_mfma_f32_32x32x2_f32 everywhere)"; DPF_F32_X9=0 python tools/conv_shape_bench.py $SH9 2>&1 | grep -v -e MIOpen -e amdgpu.ids
    echo "== DPF_G2_VEC_STORE=0 (4-byte tile stores)"; DPF_G2_VEC_STORE=0 python tools/conv_shape_bench.py $SH9 2>&1 | grep -v -e MIOpen -e amdgpu.ids
  } > "$O/${R}_conv_x9_vs_f32_per_shape.txt"
  { echo "== operand precision bf16: igemm3 NC=1 (default)"; python tools/conv_bf16_bench.py 2>&1 | grep -v -e MIOpen -e amdgpu.ids
    echo "== DPF_IGEMM3_BF=0 (igemm2 bf16 kernel)"; DPF_IGEMM3_BF=0 python tools/conv_bf16_bench.py 2>&1 | grep -v -e MIOpen -e amdgpu.ids; } > "$O/${R}_conv_bf16_per_shape.txt"
  make -C dualpixelface_amd/csrc -j8 OBJDIR=build_stamps LIB=../libdpf_hip_stamps.so EXTRA=-DDPF_STAMPS > /dev/null 2>&1
  { echo "== 16-byte tile stores"; DPF_LIB_PATH=$PWD/dualpixelface_amd/libdpf_hip_stamps.so python tools/x9_stamps.py fe32 hg32 fe32q hg64 fe96_32 2>&1 | grep -v amdgpu.ids
    echo "== DPF_G2_VEC_STORE=0"; DPF_G2_VEC_STORE=0 DPF_LIB_PATH=$PWD/dualpixelface_amd/libdpf_hip_stamps.so python tools/x9_stamps.py fe32 hg32 fe32q hg64 fe96_32 2>&1 | grep -v amdgpu.ids; } > "$O/${R}_x9_stamps.txt"
fi
cat "$O/${R}_bench.json"; cat "$O/${R}_bench_family_ms.txt"; python3 - "$O/${R}_pmc_traffic.json" <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    print('%-20s launches/step %7.1f  HBM MB/step %9.1f' % (k, v.get('launches_per_step', 0), v.get('hbm_bytes_per_step', 0) / 1e6))
PY
