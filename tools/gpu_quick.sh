python -m pytest tests/test_xattn_gpu.py tests/test_parity_gpu.py tests/test_drivers_gpu.py -m gpu -q -x -k "decode_step or generate or decode or collated or greedy" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pg -o b -- python3 tools/generate_probe.py > /tmp/gp.txt 2>&1; grep max_length /tmp/gp.txt
python tools/rocpd_stats.py /tmp/pg/b_results.db | grep -E "xdecode|hb_nt" | cut -c1-150
python tools/generate_probe.py 2>&1 | tail -2
