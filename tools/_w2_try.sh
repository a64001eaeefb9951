for n in 7 5 4 3; do
  echo "== DPF_W2_NCT=$n"
  DPF_W2_NCT=$n python tools/conv_bf16_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/  fwd.*wgrad/  wgrad/' | cut -c1-120
done
