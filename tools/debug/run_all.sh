cd /root/repo
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04_gpu_tests.txt
cat gpurun_out/r04_gpu_tests.txt
if grep -q " passed" gpurun_out/r04_gpu_tests.txt && ! grep -q "failed" gpurun_out/r04_gpu_tests.txt; then
  bash tools/gpu_profiles.sh r04 > gpurun_out/r04_profiles.log 2>&1
  tail -40 gpurun_out/r04_profiles.log
fi
