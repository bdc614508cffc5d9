"""A few launches of the noise injection (forward / backward, bf16 and float32 operands) at 2^24 outputs for rocprofv3 --pmc:
VALU instructions per output and VALU-busy share (is the bf16 kernel VALU-bound?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops
dev = "cuda"
n = 1 << 27
for dt in (torch.bfloat16, torch.float32):
    mu_r = torch.zeros(n, dtype=dt, device=dev); mu_i = torch.zeros(n, dtype=dt, device=dev)
    s2 = torch.full((n,), 0.5, dtype=dt, device=dev)
    for _ in range(4):
        ops.reparam_fwd(mu_r, mu_i, s2, None, 1, 2, inplace=True)
    for _ in range(4):
        ops.reparam_bwd(mu_r, mu_i, s2, None, 1, 2, out_dtype=dt)
    torch.cuda.synchronize()
    del mu_r, mu_i, s2
