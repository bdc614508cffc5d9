"""Complex bf16 GEMM timing for the current CPLXAMD_GEMM_VARIANT (one process per variant)."""
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
    return s.elapsed_time(e) / iters * 1e-3


dev = "cuda"
torch.manual_seed(0)
v = " ".join(f"{k[13:]}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("CPLXAMD_GEMM_"))
res = []
for (M, N, K, conj, odt) in [(8192, 4096, 4096, False, torch.bfloat16), (8192, 4096, 4096, True, torch.bfloat16),
                             (4096, 4096, 8192, True, torch.float32), (8192, 8192, 8192, False, torch.bfloat16)]:
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
    b = [torch.randn(N, K, device=dev).bfloat16() for _ in range(2)]
    out = (torch.empty(M, N, device=dev, dtype=odt), torch.empty(M, N, device=dev, dtype=odt))
    t = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, conj_b=conj, out=out))
    res.append(f"{M}x{N}x{K}{'c' if conj else ''}:{t*1e3:.3f}ms={8*M*N*K/t/1e12:.0f}TF")
# Gauss 3M (three real MFMA GEMMs + fused combine) on the headline shape, 8MNK-equivalent rate
M, N, K = 8192, 4096, 4096
a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
b = [torch.randn(N, K, device=dev).bfloat16() for _ in range(2)]
out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
for algo in (0, 1):
    t = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out, algo=algo))
    res.append(f"{'3M' if algo else '4M'}:{t*1e3:.3f}ms={8*M*N*K/t/1e12:.0f}TF(8MNK-equiv)")
del a, b, out
a, b = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
t = timeit(lambda: ops.rgemm(a, (K, 1), b, (K, 1), M, N, K))
res.append(f"real:{t*1e3:.3f}ms={2*M*N*K/t/1e12:.0f}TF")
print(f"variant {v}: " + "  ".join(res))
