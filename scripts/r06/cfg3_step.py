"""configs[2] in bf16 (CplxConv2d(64, 64, 3) + CplxBatchNorm2d, batch 256, 256 x 256, channels-last): N steps, for rocprofv3.
FOLD=0/1: the batch-norm backward apply inside the weight-gradient launch (default 1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cplxmodule_amd import Cplx, nn, conv as cv, _lib  # noqa: E402

if os.environ.get("CPLXAMD_LIB"):          # (ablation builds of the library)
    _lib.LIB_PATH = os.environ["CPLXAMD_LIB"]

cv._BN_FOLD = os.environ.get("FOLD", "1") != "0"
dev = "cuda"
torch.manual_seed(0)
layer, bn = nn.CplxConv2d(64, 64, 3).to(dev), nn.CplxBatchNorm2d(64).to(dev)
mk = lambda: (torch.randn(256, 64, 256, 256, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
              .requires_grad_(True))
x = Cplx(mk(), mk())


def step():
    layer.zero_grad(set_to_none=True); bn.zero_grad(set_to_none=True)
    x.real.grad = x.imag.grad = None
    y = bn(layer(x))
    torch.autograd.backward((y.real, y.imag), (y.real.detach(), y.imag.detach()))


for _ in range(3):
    step()
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", "10"))
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"cfg3 bf16 batch 256 fold={cv._BN_FOLD}: ms_per_step {(time.perf_counter() - t0) / n * 1e3:.3f}")
