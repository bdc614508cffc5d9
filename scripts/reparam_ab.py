"""LRT noise injection at BASELINE's cfg4 size (batch 2^20 x 2048 outputs), bf16 and fp32 I/O, for one build of the
library (CPLXAMD_LIB selects it): GB/s of algorithmic bytes, median of 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops, _lib

def med(fn, n=7):
    ts = []
    for _ in range(n + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts[2:])[n // 2] * 1e-3

dev = "cuda"
n = (1 << 20) * 2048
res = [os.path.basename(_lib.LIB_PATH)]
for dt, bf, bb in ((torch.bfloat16, 12, 10), (torch.float32, 20, 16)):
    mu_r = torch.zeros(n, dtype=dt, device=dev); mu_i = torch.zeros(n, dtype=dt, device=dev)
    s2 = torch.full((n,), 0.5, dtype=torch.float32, device=dev)
    t = med(lambda: ops.reparam_fwd(mu_r, mu_i, s2, None, 1, 2, inplace=True))
    res.append(f"{str(dt)[6:]} fwd {bf * n / t / 1e9:7.1f} GB/s ({t*1e3:.2f} ms)")
    t = med(lambda: ops.reparam_bwd(mu_r, mu_i, s2, None, 1, 2, out_dtype=dt))
    res.append(f"bwd {bb * n / t / 1e9:7.1f} GB/s ({t*1e3:.2f} ms)")
    del mu_r, mu_i, s2
print("  ".join(res))
