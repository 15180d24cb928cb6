"""Which hipBLASLt (Tensile) kernel torch.matmul picks for the training step's NT shapes — run under `rocprofv3 --kernel-trace --stats`
and read the kernel names (they encode the macro tile, wave tiling, prefetch depth and LDS path).  A yardstick only: never on the product path."""
import torch

dev = torch.device("cuda:0")
for (m, n, k) in [(47757, 2304, 768), (47757, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev).bfloat16()
    b = torch.randn(n, k, device=dev).bfloat16()
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
