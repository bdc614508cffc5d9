"""Complex bf16 GEMM (forward layout) at growing K: how much of the time is per-tile (epilogue, boundary) and how much
is the K loop.  python scripts/r02/gemm_k_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops

dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
M, N = 8192, 4096
for K in (1024, 2048, 4096, 8192, 16384):
    a = [torch.randn(M, K, device=dev).to(bf) for _ in range(2)]
    b = [(torch.randn(N, K, device=dev) * 0.01).to(bf) for _ in range(2)]
    out = (torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf))
    f = lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out)  # noqa: E731
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 10)
    t = sorted(ts)[2]
    print(f"K={K:6d}: {t:.4f} ms  {8.0 * M * N * K / t / 1e9:7.0f} TF/s  frac {8.0 * M * N * K / t / 1e9 / 2500:.3f}   per 32-deep K tile and output tile: {t * 1e3 / (K / 32) / 4:.3f} us")
