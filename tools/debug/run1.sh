cd /root/repo
echo "== x9"; timeout 900 python tools/conv_shape_bench.py --check hg32 hg64 cv64_32 fe32 fe32q fe64 fe96_32 anm96d2 off81 fe32d5 2>&1 | grep -v amdgpu
echo "== exact"; DPF_F32_X9=0 timeout 900 python tools/conv_shape_bench.py hg32 hg64 cv64_32 fe32 fe32q fe64 fe96_32 anm96d2 off81 fe32d5 2>&1 | grep -v amdgpu
