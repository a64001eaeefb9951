cd /root/repo
for v in a b c; do echo "== tile $v"; DPF_DCN_LEAN_FWD=$v timeout 600 python tools/debug/dcn_lean_check.py parity time 2>&1 | grep -v amdgpu.ids; done
for v in c; do for C in 64; do echo "== stamps tile $v C $C"; DPF_DCN_LEAN_FWD=$v DPF_LIB_PATH=/root/repo/dualpixelface_amd/libdpf_hip_stamps.so timeout 300 python tools/debug/dcn_lean_stamps.py $C 2>&1 | grep -v amdgpu.ids | head -14; done; done
