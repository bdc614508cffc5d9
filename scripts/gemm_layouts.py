"""Complex bf16 GEMM, forward shape, by operand layout: B as [N, K] (K contiguous) vs [K, N] (N contiguous)."""
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
a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
b = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(2)]
bt = [t.t().contiguous() for t in b]
at = [t.t().contiguous() for t in a]
out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
bias = (torch.zeros(N, device=dev), torch.zeros(N, device=dev))
for rep in range(2):
    for conj in (False, True):
        t_nn = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, bias=bias, conj_b=conj, out=out))
        t_nt = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), bt[0], bt[1], (1, N), M, N, K, bias=bias, conj_b=conj, out=out))
        t_tt = timeit(lambda: ops.cgemm(at[0], at[1], (1, M), bt[0], bt[1], (1, N), M, N, K, bias=bias, conj_b=conj, out=out))
        print(f"conj={conj}: A[M,K] B[N,K] {t_nn:.3f} ms | A[M,K] B[K,N] {t_nt:.3f} ms | A[K,M] B[K,N] {t_tt:.3f} ms")
