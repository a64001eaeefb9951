#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
for e in "A=1" "DPF_IGEMM3_RSTEP=0" "DPF_F32_X9=0" "DPF_IGEMM3=0" "DPF_FEATURES_TWO_STREAMS=0 DPF_WGRAD_ASYNC=0"; do
echo "== $e"; for t in train_32x48_b2 train_64x96_b1; do env $e python tools/debug/grad_fp64_measure.py $t 2>&1 | grep train_; done
done
