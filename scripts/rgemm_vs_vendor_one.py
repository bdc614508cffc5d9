"""A few launches of our real bf16 GEMM and of the vendor's on the same shape (for --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops
dev = "cuda"
torch.manual_seed(0)
M, N, K = 8192, 4096, 4096
a = torch.randn(M, K, device=dev).bfloat16()
b = torch.randn(N, K, device=dev).bfloat16()
n = int(os.environ.get("ITERS", "12"))
for which in ("ours", "vendor", "ours", "vendor"):
    for _ in range(n):
        if which == "ours":
            ops.rgemm(a, (K, 1), b, (K, 1), M, N, K, out_dtype=torch.bfloat16)
        else:
            torch.matmul(a, b.t())
    torch.cuda.synchronize()
