set -x
OUT=gpurun_out/r03e
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -x -q --durations=8 -k "gemm_nt or attention or oracle_tokens" ) > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
LAKO_PROBE_TOKENS=47757 python tools/bench_ops.py --only gemm --variants 2,7 --iters 20 > $OUT/bench_ops_47757_v27.log 2>&1 || true
python - <<'PY' > $OUT/bench_ops_192.txt 2>&1
import os, sys, torch
sys.path.insert(0, os.getcwd())
from lako_amd.ops import HipOps
ops = HipOps(); dev = torch.device("cuda:0"); T = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 47757
for (N, K, kw) in [(768, 768, "plain"), (768, 768, "resid"), (768, 3072, "plain"), (768, 3072, "resid"), (768, 2304, "plain"), (2304, 768, "plain"), (3072, 768, "relu"), (3072, 768, "aux")]:
    A = torch.randn(M, K, device=dev).to(T); B = torch.randn(N, K, device=dev).to(T); C = torch.empty(M, N, dtype=T, device=dev)
    R = torch.randn(M, N, device=dev).to(T)
    k2 = {"plain": {}, "resid": dict(resid=R, drop=(0.1, 1, 2)), "relu": dict(relu=True, drop=(0.1, 1, 2)), "aux": dict(aux=R, aux_scale=1.1)}[kw]
    res = []
    for t192, var in ((0, -1), (1, -1), (1, 7), (1, 2)):
        ops.set_tuning("gemm_nt_tile192", t192); ops.set_tuning("gemm_nt_variant", var)
        res.append(timeit(lambda: ops.gemm_nt(A, B, C, **k2)))
    ops.set_tuning("gemm_nt_variant", -1)
    fl = 2.0 * M * N * K
    print(f"[{M},{K}]x[{N},{K}] {kw:6s}: auto without 192-row tiles {res[0]:7.1f} us ({fl/res[0]/1e6:6.0f} TF/s) | auto {res[1]:7.1f} us ({fl/res[1]/1e6:6.0f}) | forced 192x256 {res[2]:7.1f} | forced 256x256 (no tail split) {res[3]:7.1f}", flush=True)
PY
cat $OUT/bench_ops_192.txt
python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; head -12 $OUT/bench.err; tail -1 $OUT/bench.err
LAKO_TUNING=gemm_nt_tile192=0 python bench.py --no-cpu-baseline --all-valid-steps 0 --steps 20 --warmup 5 --breakdown > $OUT/bench_no192.json 2> $OUT/bench_no192.err
cut -c1-200 $OUT/bench_no192.json; head -4 $OUT/bench_no192.err
