import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lako_amd.ops import HipOps
ops = HipOps(); dev = torch.device("cuda:0"); T = torch.bfloat16
Me, d, f = 64000, 768, 3072
def timeit(name, fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / iters:9.1f} us", flush=True)
A, B = torch.randn(Me, d, device=dev).to(T), torch.randn(f, d, device=dev).to(T)
C = torch.empty(Me, f, dtype=T, device=dev)
R = torch.randn(Me, f, device=dev).to(T)
for _ in range(2):
    timeit("wi plain", lambda: ops.gemm_nt(A, B, C))
    timeit("wi relu", lambda: ops.gemm_nt(A, B, C, relu=True))
    timeit("wi relu+drop", lambda: ops.gemm_nt(A, B, C, relu=True, drop=(0.1, 1, 2)))
    timeit("wi drop", lambda: ops.gemm_nt(A, B, C, drop=(0.1, 1, 2)))
    timeit("wi resid", lambda: ops.gemm_nt(A, B, C, resid=R))
    timeit("wi resid+drop", lambda: ops.gemm_nt(A, B, C, resid=R, drop=(0.1, 1, 2)))
    timeit("wi aux", lambda: ops.gemm_nt(A, B, C, aux=R, aux_scale=1.1))
