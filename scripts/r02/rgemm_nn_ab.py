"""Real bf16 (N,N) GEMM of the variance path, bf16 out vs float32 out, persistent kernel on / off (env
CPLXAMD_GEMM_PERSIST is read once per process: run twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops


def timeit(fn, iters=40, warm=15):
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
out = []
for B, I, O in ((8192, 4096, 4096), (1 << 18, 2048, 2048)):
    a = torch.randn(B, I, device=dev).square().bfloat16()
    S = torch.randn(O, I, device=dev).exp().bfloat16()
    for dt in (torch.bfloat16, torch.float32):
        c = torch.empty(B, O, device=dev, dtype=dt)
        t = timeit(lambda: ops.rgemm(a, (I, 1), S, (I, 1), B, O, I, out=c))
        out.append(f"{B}x{I}x{O} {str(dt)[6:]}: {t*1e3:.3f} ms {2*B*I*O/t/1e12:.0f} TF")
print("persist=" + os.environ.get("CPLXAMD_GEMM_PERSIST", "1"), " | ".join(out))
