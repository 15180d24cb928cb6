python -m pytest tests/test_xattn_gpu.py -m gpu -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/px -o b -- python3 tools/xattn_time.py > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/px/b_results.db | grep -E "contract" | cut -c1-40,95-160
XB=8 XKEYS=8000 XH=16 XZ=4 rocprofv3 --kernel-trace --stats -d /tmp/px2 -o b -- python3 tools/xattn_time.py > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/px2/b_results.db | grep -E "contract|xscores|xcontext" | cut -c1-40,85-160
