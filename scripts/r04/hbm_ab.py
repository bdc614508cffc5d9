"""HBM-bound kernels at the sizes BASELINE.json quotes, one library build per process (CPLXAMD_LIB=<lib.so>):
LRT noise injection at 2^20 x 2048 outputs (bf16, variance bf16 / float32), the in-step backward with bias sums
(8192 x 4096), KL on a 16384^2 complex weight.  GB/s of algorithmic bytes, HIP events, median of 7."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd import ops  # noqa: E402

dev = "cuda"


def med(fn, n=7):
    ts = []
    for _ in range(n + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts[2:])[n // 2] * 1e-3


out = {}
n = (1 << 20) * 2048
bf = torch.bfloat16
mu_r = torch.randn(1 << 20, device=dev).to(bf).repeat(2048)
mu_i = torch.randn(1 << 20, device=dev).to(bf).repeat(2048)
s2f = torch.rand(1 << 20, device=dev).repeat(2048)
s2 = s2f.to(bf)
out["fwd 10B s2bf16"] = 10 * n / med(lambda: ops.reparam_fwd(mu_r, mu_i, s2, None, 1, 2, inplace=True)) / 1e9
out["bwd 8B s2bf16"] = 8 * n / med(lambda: ops.reparam_bwd(mu_r, mu_i, s2, None, 1, 2, out_dtype=bf)) / 1e9
out["fwd 12B s2f32"] = 12 * n / med(lambda: ops.reparam_fwd(mu_r, mu_i, s2f, None, 1, 2, inplace=True)) / 1e9
out["bwd 10B s2f32"] = 10 * n / med(lambda: ops.reparam_bwd(mu_r, mu_i, s2f, None, 1, 2, out_dtype=bf)) / 1e9
del mu_r, mu_i, s2, s2f
B, O = 8192, 4096
gr, gi = torch.randn(B, O, device=dev).to(bf), torch.randn(B, O, device=dev).to(bf)
s2 = torch.rand(B, O, device=dev).to(bf)
br, bi = torch.empty(O, device=dev), torch.empty(O, device=dev)
t = med(lambda: ops.reparam_bwd(gr, gi, s2, None, 1, 2, out_dtype=bf, bias_sums=(B, O, (br, bi))), 15)
out["bwd_cols 8192x4096 8B"] = 8 * B * O / t / 1e9
t = med(lambda: ops.reparam_fwd(gr, gi, s2, None, 1, 2, inplace=True), 15)
out["fwd 8192x4096 10B"] = 10 * B * O / t / 1e9
del gr, gi, s2
m = 16384 * 16384
wr = torch.randn(m, device=dev) * 0.01
wi = torch.randn(m, device=dev) * 0.01
ls2 = torch.empty(m, device=dev).uniform_(-12, 4)
out["kl_fwd 12B"] = 12 * m / med(lambda: ops.kl_fwd("cplx_vd", wr, wi, ls2)) / 1e9
out["kl_fwd_bwd 24B"] = 24 * m / med(lambda: ops.kl_fwd_bwd("cplx_vd", wr, wi, ls2)) / 1e9
print(os.environ.get("CPLXAMD_LIB", "prod").split("libcplxamd")[-1].ljust(12) + "  ".join(f"{k}: {v:7.0f}" for k, v in out.items()), flush=True)
