"""Batch-norm row kernels alone on configs[2]'s planes (rows = 256 * 254 * 254, 64 channels, bf16 channels-last): forward
and backward of the layer with nothing in front of it, per library build (CPLXAMD_LIB; BN_NT = 0..3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cplxmodule_amd import Cplx, nn, _lib  # noqa: E402

if os.environ.get("CPLXAMD_LIB"):
    _lib.LIB_PATH = os.environ["CPLXAMD_LIB"]
dev = "cuda"
bn = nn.CplxBatchNorm2d(64).to(dev)
DT = torch.float32 if os.environ.get("DT") == "f32" else torch.bfloat16
mk = lambda: (torch.randn(256, 64, 254, 254, device=dev).to(DT).contiguous(memory_format=torch.channels_last)  # noqa: E731
              .requires_grad_(True))
x = Cplx(mk(), mk())
g = (mk().detach(), mk().detach())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(13):
    x.real.grad = x.imag.grad = None
    bn.zero_grad(set_to_none=True)
    ev[0].record()
    y = bn(x)
    ev[1].record()
    torch.autograd.backward((y.real, y.imag), g)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 3:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"{os.path.basename(_lib.LIB_PATH)}: bn forward {tf / 10:.3f} ms (2 + 2 + 2 planes), backward {tb / 10:.3f} ms (4 + 4 + 2 planes)")
