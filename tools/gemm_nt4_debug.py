import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps
dev = torch.device("cuda:0")
ops = HipOps()
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 512)))
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
C2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); C9 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
ops.set_tuning("gemm_nt_variant", 2); ops.gemm_nt(A, B, C2)
ops.set_tuning("gemm_nt_variant", 9); ops.gemm_nt(A, B, C9)
torch.cuda.synchronize()
ref = (A.float() @ B.float().t())
bad = (C2.view(torch.int16) != C9.view(torch.int16))
print("mismatch", int(bad.sum()), "of", M * N, "nan in v9", int(torch.isnan(C9).sum()))
print("max |v9-ref|", float((C9.float() - ref).abs().max()), "max |v2-ref|", float((C2.float() - ref).abs().max()))
idx = bad.nonzero()
if len(idx):
    r, c = idx[:, 0], idx[:, 1]
    print("rows mod 256 hist (16-bins):", torch.bincount((r % 256) // 16, minlength=16).tolist())
    print("cols mod 256 hist (16-bins):", torch.bincount((c % 256) // 16, minlength=16).tolist())
    print("rows mod 16:", torch.bincount(r % 16, minlength=16).tolist())
    print("cols mod 16:", torch.bincount(c % 16, minlength=16).tolist())
    d = (C9.float() - C2.float())[bad]
    print("diff abs mean", float(d.abs().mean()), "max", float(d.abs().max()), "ref scale", float(ref.abs().mean()))
    print("first few", idx[:10].tolist())
