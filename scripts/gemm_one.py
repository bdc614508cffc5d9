"""One complex bf16 GEMM shape, a few launches (for rocprofv3 --pmc passes and DVFS checks).
env: ZERO=1 -> zero-filled operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops
dev = "cuda"
torch.manual_seed(0)
M, N, K = 8192, 4096, 4096
zero = os.environ.get("ZERO") == "1"
mk = (lambda *s: torch.zeros(*s, device=dev).bfloat16()) if zero else (lambda *s: torch.randn(*s, device=dev).bfloat16())
a = [mk(M, K) for _ in range(2)]
b = [mk(N, K) for _ in range(2)]
out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
n = int(os.environ.get("ITERS", "30"))
for _ in range(10):
    ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / n * 1e-3
print(f"variant {os.environ.get('CPLXAMD_GEMM_VARIANT','dflt')} zero={zero}: {t*1e3:.3f} ms {8*M*N*K/t/1e12:.0f} TF/s")
