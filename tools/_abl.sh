for d in 0 1 2 4 8 16 12 28 30 31; do echo "DBG=$d"; DPF_DBG=$d python tools/dcn_bench.py all 64 2>&1 | grep "C="; done
