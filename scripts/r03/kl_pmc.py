"""A few launches of the complex-VD KL kernels on a 8192^2 weight for rocprofv3 --pmc (VALU instructions per element, VALU busy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops
dev = "cuda"
m = 8192 * 8192
wr = torch.randn(m, device=dev) * 0.01
wi = torch.randn(m, device=dev) * 0.01
ls2 = torch.empty(m, device=dev).uniform_(-12, 4)
for _ in range(4):
    ops.kl_fwd("cplx_vd", wr, wi, ls2)
for _ in range(4):
    ops.kl_fwd_bwd("cplx_vd", wr, wi, ls2)
for _ in range(4):
    ops.prep_kl("cplx_vd", wr, wi, ls2, True)
for _ in range(4):
    ops.kl_fwd_bwd("cplx_ard", wr, wi, ls2)
torch.cuda.synchronize()
