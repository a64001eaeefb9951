#!/bin/bash
cd /root/repo
make -C dualpixelface_amd/csrc -j8 OBJDIR=build_stamps LIB=../libdpf_hip_stamps.so EXTRA=-DDPF_STAMPS > /tmp/mk.log 2>&1 || tail -20 /tmp/mk.log
DPF_LIB_PATH=/root/repo/dualpixelface_amd/libdpf_hip_stamps.so timeout 600 python tools/debug/x9_stamps.py fe32 hg32 fe32q hg64 fe96_32 2>&1 | grep -v amdgpu.ids
