import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps
dev = torch.device("cuda:0"); ops = HipOps()
if os.environ.get("TNBIG"): ops.set_tuning("gemm_tn_big", int(os.environ["TNBIG"]))
K, M, N = (int(x) for x in sys.argv[1:4]); split = int(sys.argv[4]) if len(sys.argv) > 4 else -2
A = (torch.randn(K, M, device=dev) * 0.5).bfloat16(); B = (torch.randn(K, N, device=dev) * 0.5).bfloat16()
C0, C1 = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
ops.set_tuning("gemm_tn_four", 0); ops.gemm_tn(A, B, C0, split_k=split); torch.cuda.synchronize(); print("old ok", flush=True)
ops.set_tuning("gemm_tn_four", 1); ops.gemm_tn(A, B, C1, split_k=split); torch.cuda.synchronize(); print("new ok", flush=True)
ref = A.float().t() @ B.float()
print("equal", torch.equal(C0, C1), "err old", float((C0 - ref).abs().max() / ref.abs().max()), "err new", float((C1 - ref).abs().max() / ref.abs().max()))
bad = (C0 != C1).nonzero()
if len(bad): print("mismatches", len(bad), bad[:8].tolist(), "rows%256 hist", torch.bincount((bad[:, 0] % 256) // 16, minlength=16).tolist(), "cols%256", torch.bincount((bad[:, 1] % 256) // 16, minlength=16).tolist())
nf = (K // 64) * 64
Af, Bf = A.float(), B.float()
P = Af[nf:].t() @ Bf[nf:]
full = Af[:nf].t() @ Bf[:nf]
def rel(x): return float(x.abs().max() / ref.abs().max())
print("C1 - ref:", rel(C1 - ref), " C1 - (ref - P):", rel(C1 - (ref - P)), " C1 - full:", rel(C1 - full), " C1 - P:", rel(C1 - P), "finite", bool(torch.isfinite(C1).all()))
for s in range(0, min(nf, 256), 64):
    Fs = Af[s:s + 64].t() @ Bf[s:s + 64]
    print(f"  C1 - (ref - step@{s}):", rel(C1 - (ref - Fs)), f" C1 - (ref + step@{s}):", rel(C1 - (ref + Fs)))
