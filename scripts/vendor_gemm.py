"""Calibration: the vendor bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the bench shapes, next to
the hand-written kernels.  Not used by the product; numbers go to profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops


def timeit(fn, iters=30, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


dev = "cuda"
torch.manual_seed(0)
M, N, K = 8192, 4096, 4096
ar, ai = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
br, bi = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(2)]
out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
t = timeit(lambda: torch.matmul(ar, br.t()))
print(f"vendor real bf16 {M}x{N}x{K}: {t:.3f} ms = {2*M*N*K/t/1e9:.0f} TF/s")
t = timeit(lambda: ops.rgemm(ar, (K, 1), br, (K, 1), M, N, K, out_dtype=torch.bfloat16))
print(f"ours   real bf16 {M}x{N}x{K}: {t:.3f} ms = {2*M*N*K/t/1e9:.0f} TF/s")


def vendor_cplx():
    re = torch.matmul(ar, br.t()) - torch.matmul(ai, bi.t())
    im = torch.matmul(ar, bi.t()) + torch.matmul(ai, br.t())
    return re, im


t = timeit(vendor_cplx)
print(f"vendor complex (4 GEMMs + 2 adds, the reference's formulation): {t:.3f} ms = {8*M*N*K/t/1e9:.0f} TF/s")
cat_a = torch.cat([ar, ai], 1)
cat_b = torch.cat([torch.cat([br, -bi], 1), torch.cat([bi, br], 1)], 0)
t = timeit(lambda: torch.matmul(cat_a, cat_b.t()))
print(f"vendor complex (one [M,2K] x [2N,2K]^T GEMM, operands pre-concatenated): {t:.3f} ms = {8*M*N*K/t/1e9:.0f} TF/s")
t = timeit(lambda: ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, N, K, out=out))
print(f"ours   complex (fused 4M kernel): {t:.3f} ms = {8*M*N*K/t/1e9:.0f} TF/s")
