"""The forward launch of the headline step -- complex (N,N) GEMM, 8192 x 4096 x 4096, bf16 -- 14 times, for rocprofv3 passes
(scripts/r06/clock_vs_fetch.sh).  TOUCH=1: a read of both weight planes (2 x 32 MiB) right before every launch, so that B
sits in the Infinity Cache when the kernel starts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cplxmodule_amd import ops  # noqa: E402

B, I, O = 8192, 4096, 4096
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
bound = (1.0 / (2 * I)) ** 0.5
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.empty(O, I, device=dev).uniform_(-bound, bound).to(bf) for _ in range(2))
touch = os.environ.get("TOUCH", "0") == "1"
which = os.environ.get("LAYOUT", "NN")
gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
sink = torch.zeros((), device=dev)
for _ in range(14):
    if touch:
        sink += wr.view(torch.int16).sum() + wi.view(torch.int16).sum()
    if which == "NN":
        ops.cgemm(xr, xi, (I, 1), wr, wi, (I, 1), B, O, I, out_dtype=bf)
    else:   # the input gradient: G conj(W), W read K-major as stored
        ops.cgemm(gr, gi, (O, 1), wr, wi, (1, I), B, I, O, conj_b=True, out_dtype=bf)
torch.cuda.synchronize()
