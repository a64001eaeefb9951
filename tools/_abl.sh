for d in 0 2 8 16 24 26 31; do echo "DBG=$d"; DPF_DBG=$d python tools/dcn_bench.py all 64 2>&1 | grep "C="; done
