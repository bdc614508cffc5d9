"""BASELINE configs[2] (CplxConv2d(64, 64, 3) on 256 x 256 + CplxBatchNorm2d, bf16, batch 256, channels-last, fwd + bwd) with
and without the batch-norm moments in the convolution's epilogue (CPLXAMD_CONV_BN_MOMENTS=0|1, one process each):
ms per step and the forward kernels' share (HIP events around the two module calls)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd import Cplx  # noqa: E402
from cplxmodule_amd.nn.modules.batchnorm import CplxBatchNorm2d  # noqa: E402
from cplxmodule_amd.nn.modules.conv import CplxConv2d  # noqa: E402

dev = "cuda"
B = int(os.environ.get("BATCH", "256"))
bf = torch.bfloat16
cl = torch.channels_last
torch.manual_seed(0)
convl, bn = CplxConv2d(64, 64, 3).to(dev), CplxBatchNorm2d(64).to(dev)
x = Cplx(torch.randn(B, 64, 256, 256, device=dev).to(bf).contiguous(memory_format=cl).requires_grad_(True),
         torch.randn(B, 64, 256, 256, device=dev).to(bf).contiguous(memory_format=cl).requires_grad_(True))
gy = torch.randn(B, 64, 254, 254, device=dev).to(bf).contiguous(memory_format=cl)
tf, tb, tc = [], [], []
for it in range(9):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    convl.zero_grad(set_to_none=True); bn.zero_grad(set_to_none=True)
    x.real.grad = x.imag.grad = None
    e[0].record()
    h = convl(x)
    e[1].record()
    y = bn(h)
    e[2].record()
    torch.autograd.backward((y.real, y.imag), (gy, gy))
    e[3].record()
    torch.cuda.synchronize()
    tc.append(e[0].elapsed_time(e[1])); tf.append(e[1].elapsed_time(e[2])); tb.append(e[2].elapsed_time(e[3]))
    del h, y
med = lambda t: sorted(t[3:])[len(t[3:]) // 2]  # noqa: E731
print(f"moments={os.environ.get('CPLXAMD_CONV_BN_MOMENTS', '1')}  conv fwd {med(tc):6.3f}  bn fwd {med(tf):6.3f}  backward {med(tb):6.3f}  "
      f"step {med(tc) + med(tf) + med(tb):6.3f} ms")
