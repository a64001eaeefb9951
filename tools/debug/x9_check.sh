#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
for e in "A=1" "DPF_FEATURES_TWO_STREAMS=0 DPF_WGRAD_ASYNC=0" "DPF_F32_X9=0" "DPF_IGEMM3_RSTEP=0" "DPF_IGEMM3=0" "DPF_G2_VEC_STORE=0 DPF_IGEMM3_SH=0"; do
echo "== $e"; env $e python tools/debug/grad_rel_measure.py 2>&1 | grep train_
done
