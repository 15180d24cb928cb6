mkdir -p gpurun_out/r02j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o b -- python3 bench.py --model large --n-passages 100 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --all-valid-steps 0 > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/pc5/b_results.db > gpurun_out/r02j/c5_kernel_stats.csv; head -24 gpurun_out/r02j/c5_kernel_stats.csv | awk -F, '{n=$1; if (length(n)>60) n=substr(n,1,60); print n, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}'
python tools/rocpd_timeline.py /tmp/pc5/b_results.db 3 | head -3
