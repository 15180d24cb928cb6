for d in 0 1024 2048 3072 4096 5120; do echo "DBG=$d"; LAKO_LIB=$PWD/lako_amd/liblako_hip_exp.so LAKO_ATTN_DEBUG=$d python tools/attn_time.py 2>&1 | tail -1; done
