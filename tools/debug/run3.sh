cd /root/repo
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for i in 1 2 3; do timeout 600 python tools/parity_probe.py 2>&1 | grep -v amdgpu.ids | grep "relL2\|loss rel\|median" ; done
