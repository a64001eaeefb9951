cd /root/repo
echo "== lean"; timeout 900 python tools/debug/dcn_bwd_check.py parity time 2>&1 | grep -v amdgpu.ids
echo "== old offset kernel"; DPF_DCN_LEAN=5 timeout 600 python tools/debug/dcn_bwd_check.py time 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "headline" -s 2>&1 | grep -v amdgpu.ids | tail -8
