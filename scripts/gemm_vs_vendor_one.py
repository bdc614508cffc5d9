"""A few launches of our complex GEMM and of the vendor's real GEMM of the same flop count (the
concatenated formulation), for rocprofv3 --pmc passes (clock and MFMA-busy comparison)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops
dev = "cuda"
torch.manual_seed(0)
M, N, K = 8192, 4096, 4096
a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
b = [torch.randn(N, K, device=dev).bfloat16() for _ in range(2)]
out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
ca = torch.cat(a, 1)
cb = torch.cat([torch.cat([b[0], -b[1]], 1), torch.cat([b[1], b[0]], 1)], 0)
n = int(os.environ.get("ITERS", "12"))
for which in ("ours", "vendor", "ours", "vendor"):
    for _ in range(n):
        if which == "ours":
            ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out)
        else:
            torch.matmul(ca, cb.t())
    torch.cuda.synchronize()
