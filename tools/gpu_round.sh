set -x
mkdir -p gpurun_out/r02c
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r02c/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02c/pytest.log
( time python bench.py ) > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
python bench.py --breakdown --no-cpu-baseline --all-valid-steps 0 --steps 10 --warmup 5 > gpurun_out/r02c/bench_breakdown.json 2> gpurun_out/r02c/bench_breakdown.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r02c/prof -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-valid-steps 0 > gpurun_out/r02c/bench_under_rocprof.json 2> gpurun_out/r02c/prof.log
ls -R gpurun_out/r02c/prof | head -20
tools/run_pmc.sh gpurun_out/r02c/pmc_attn tools/attn_probe.py sq1 sq2 mfma
LAKO_PROBE_TOKENS=48000 tools/run_pmc.sh gpurun_out/r02c/pmc_gemm tools/gemm_probe.py mfma
python tools/pmc_summary.py gpurun_out/r02c/pmc_attn --match attn_,enc_ > gpurun_out/r02c/attn_pmc.txt
python tools/pmc_summary.py gpurun_out/r02c/pmc_gemm --match gemm_ > gpurun_out/r02c/gemm_pmc.txt
# keep the merged output small: drop the raw per-dispatch CSVs of the PMC passes
find gpurun_out/r02c/pmc_attn gpurun_out/r02c/pmc_gemm -name "*.csv" -size +2M -delete
tail -5 gpurun_out/r02c/pytest.log; cat gpurun_out/r02c/bench.json
