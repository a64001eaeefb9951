#!/bin/bash
# Collects the round-2 evidence under gpurun_out/ (copied to profiles/ afterwards): bench lines, per-shape table, rocprofv3 kernel
# stats, PMC traffic (separate FETCH / WRITE passes) and SQ counters of the dominant kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
python bench.py --shapes gpurun_out/r02/r02_conv_shape_table.txt > gpurun_out/r02/r02_bench.json 2> gpurun_out/r02/bench.err
python bench.py --height 512 --width 768 --no-cpu-baseline > gpurun_out/r02/r02_bench_c2_512x768.json 2>> gpurun_out/r02/bench.err
python bench.py --model psmnet --batch 2 --no-cpu-baseline > gpurun_out/r02/r02_bench_c4_psmnet_train.json 2>> gpurun_out/r02/bench.err
python bench.py --precision bf16 --no-cpu-baseline --shapes gpurun_out/r02/r02_conv_shape_table_bf16.txt > gpurun_out/r02/r02_bench_c5_bf16_b4.json 2>> gpurun_out/r02/bench.err
python bench.py --precision bf16-2d --no-cpu-baseline > gpurun_out/r02/r02_bench_c5_bf16_2d_only.json 2>> gpurun_out/r02/bench.err
python bench.py --wgrad-async --no-cpu-baseline > gpurun_out/r02/r02_bench_wgrad_async.json 2>> gpurun_out/r02/bench.err
python bench.py --precision bf16 --wgrad-async --no-cpu-baseline > gpurun_out/r02/r02_bench_c5_bf16_wgrad_async.json 2>> gpurun_out/r02/bench.err
python bench.py --model psmnet --batch 2 --precision bf16 --no-cpu-baseline > gpurun_out/r02/r02_bench_psmnet_bf16.json 2>> gpurun_out/r02/bench.err
python bench.py --model nnet --batch 2 --precision bf16 --no-cpu-baseline > gpurun_out/r02/r02_bench_nnet_bf16.json 2>> gpurun_out/r02/bench.err
python bench.py --model stereonet --precision bf16 --no-cpu-baseline > gpurun_out/r02/r02_bench_stereonet_bf16.json 2>> gpurun_out/r02/bench.err
python bench.py --height 512 --width 768 --precision bf16 --no-cpu-baseline > gpurun_out/r02/r02_bench_c2_512x768_bf16.json 2>> gpurun_out/r02/bench.err
python tools/conv_bf16_bench.py > gpurun_out/r02/r02_conv_bf16_vs_f32_per_shape.txt 2>> gpurun_out/r02/bench.err
python bench.py --workload cost_volume --no-cpu-baseline > gpurun_out/r02/r02_bench_cost_volume_stage.json 2>> gpurun_out/r02/bench.err
python bench.py --workload cost_volume_fix --no-cpu-baseline > gpurun_out/r02/r02_bench_cost_volume_fix_stage.json 2>> gpurun_out/r02/bench.err
python bench.py --model nnet --batch 2 --no-cpu-baseline > gpurun_out/r02/r02_bench_nnet_train.json 2>> gpurun_out/r02/bench.err
python bench.py --model stereonet --no-cpu-baseline > gpurun_out/r02/r02_bench_stereonet_train.json 2>> gpurun_out/r02/bench.err
python tools/facedp_bench.py --out gpurun_out/r02/r02_facedp_bench.json > /dev/null 2>> gpurun_out/r02/bench.err
python bench.py --workload psm_volume --batch 2 --no-cpu-baseline > gpurun_out/r02/r02_bench_psm_volume.json 2>> gpurun_out/r02/bench.err
bash tools/gpu_prof.sh r02 > gpurun_out/r02/r02_bench_family_ms.txt 2>&1
cp gpurun_out/prof_r02_kernel_stats.csv gpurun_out/r02/r02_bench_kernel_stats.csv
bash tools/gpu_prof.sh r02bf16 --precision bf16 > gpurun_out/r02/r02_bench_c5_bf16_family_ms.txt 2>&1
cp gpurun_out/prof_r02bf16_kernel_stats.csv gpurun_out/r02/r02_bench_c5_bf16_kernel_stats.csv
rm -rf gpurun_out/pmc_f2 gpurun_out/pmc_w2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f2 -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w2 -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f2/p_counter_collection.csv gpurun_out/pmc_w2/p_counter_collection.csv gpurun_out/r02/r02_pmc_traffic.json
rm -rf gpurun_out/pmc_f3 gpurun_out/pmc_w3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f3 -o p -- python3 bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w3 -o p -- python3 bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f3/p_counter_collection.csv gpurun_out/pmc_w3/p_counter_collection.csv gpurun_out/r02/r02_pmc_traffic_bf16.json
export PMC_FILTER="igemm2 wgrad2 dcn_ pointwise"
bash tools/gpu_pmc.sh r02sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02/r02_sq_counters.txt 2>&1
bash tools/gpu_pmc.sh r02grbm GRBM_GUI_ACTIVE -- bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02/r02_grbm_cycles.txt 2>&1
export PMC_FILTER="igemm2 wgrad2"
bash tools/gpu_pmc.sh r02sqbf SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02/r02_sq_counters_bf16.txt 2>&1
bash tools/gpu_pmc.sh r02grbmbf GRBM_GUI_ACTIVE -- bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02/r02_grbm_cycles_bf16.txt 2>&1
python bench.py --cpu-baseline-only 512x768:16,64 > gpurun_out/r02/r02_cpu_baseline_c2.txt 2>> gpurun_out/r02/bench.err
cat gpurun_out/r02/r02_bench.json; cat gpurun_out/r02/r02_bench_family_ms.txt; cat gpurun_out/r02/r02_pmc_traffic.json | head -40; cat gpurun_out/r02/r02_cpu_baseline_c2.txt
