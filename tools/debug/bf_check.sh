#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
echo "== igemm3 bf"; timeout 600 python tools/conv_bf16_bench.py 2>&1 | grep -v MIOpen | tail -14
echo "== igemm2 bf"; DPF_IGEMM3_BF=0 timeout 600 python tools/conv_bf16_bench.py 2>&1 | grep -v MIOpen | tail -14
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -k "bf16" 2>&1 | tail -4
echo "== bench bf16"; timeout 600 python bench.py --precision bf16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
