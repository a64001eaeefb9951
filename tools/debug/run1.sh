cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fallbacks.py -m gpu -x -q 2>&1 | tail -5
