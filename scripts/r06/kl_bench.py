"""KL kernels at 16384^2 / 8192^2 (complex VD): GB/s of the fused forward + backward (24 B/elt) and the forward (12 B/elt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cplxmodule_amd import ops  # noqa: E402


def med(fn, n=9):
    ts = []
    for _ in range(n + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts[2:])[n // 2] * 1e-3


for side in (16384, 8192):
    m = side * side
    dev = "cuda"
    torch.manual_seed(0)
    wr, wi = torch.randn(m, device=dev) * 0.01, torch.randn(m, device=dev) * 0.01
    ls2 = torch.empty(m, device=dev).uniform_(-12, 4)
    for kind in ("cplx_vd", "cplx_ard"):
        t = med(lambda: ops.kl_fwd(kind, wr, wi, ls2))
        t2 = med(lambda: ops.kl_fwd_bwd(kind, wr, wi, ls2))
        print(f"{side}^2 {kind}: kl_fwd {12 * m / t / 1e9:7.1f} GB/s ({t * 1e6:7.1f} us)   kl_fwd_bwd {24 * m / t2 / 1e9:7.1f} GB/s ({t2 * 1e6:7.1f} us)", flush=True)
    del wr, wi, ls2
    torch.cuda.empty_cache()
