python -m pytest tests/test_xattn_gpu.py tests/test_real_size_gpu.py -m gpu -q -x -k "softmax or reassoc or encoder_space or cross_attention_20000 or c4_batch1_bf16" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/px -o b -- python3 tools/xattn_time.py > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/px/b_results.db | grep -E "softmax" | awk -F, '{print substr($1,1,70), $2, $4, $6, $7}'
