cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "deform or anm or full_size" 2>&1 | tail -5
timeout 600 python tools/debug/dcn_bwd_check.py parity 2>&1 | grep -v amdgpu.ids
