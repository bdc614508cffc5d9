"""Real bf16 GEMM in the three layouts of the LRT variance path (fwd NN, dgrad NT, wgrad TT)."""
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
B, I, O = 8192, 4096, 4096
a = torch.randn(B, I, device=dev).square().bfloat16()
S = torch.randn(O, I, device=dev).exp().bfloat16()
g = torch.randn(B, O, device=dev).bfloat16()
res = []
t = timeit(lambda: ops.rgemm(a, (I, 1), S, (I, 1), B, O, I))
res.append(f"fwd NN {t*1e3:.3f} ms {2*B*I*O/t/1e12:.0f} TF")
t = timeit(lambda: ops.rgemm(g, (O, 1), S, (1, I), B, I, O, out_dtype=torch.bfloat16))
res.append(f"dgrad NT {t*1e3:.3f} ms {2*B*I*O/t/1e12:.0f} TF")
t = timeit(lambda: ops.rgemm(g, (1, O), a, (1, I), O, I, B))
res.append(f"wgrad TT {t*1e3:.3f} ms {2*B*I*O/t/1e12:.0f} TF")
print(os.environ.get("CPLXAMD_LIB", "default"), " | ".join(res))
