"""CplxConv2dVD(64, 64, 3, padding 1) on 256 x 256 bf16 images, training-mode forward (mean conv + variance conv + noise
injection) + KL + backward: channels-last kernels vs the planar (round-1) path.   python scripts/lrt_conv_bench.py [B=32] [cl]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import Cplx, conv  # noqa: E402
from cplxmodule_amd.nn import relevance as rel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda"


def run(cl):
    conv._CL_ENABLED = cl
    torch.manual_seed(0)
    layer = rel.CplxConv2dVD(64, 64, 3, padding=1).to(dev)
    fmt = torch.channels_last if cl else torch.contiguous_format
    mk = lambda: torch.randn(B, 64, 256, 256, device=dev).bfloat16().contiguous(memory_format=fmt).requires_grad_(True)  # noqa: E731
    x = Cplx(mk(), mk())
    klw = torch.tensor(1e-3, device=dev)

    def step():
        layer.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = layer(x)
        kl = sum(rel.penalties(layer))
        torch.autograd.backward((y.real, y.imag, kl), (y.real.detach(), y.imag.detach(), klw))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20


modes = (True,) if len(sys.argv) > 2 and sys.argv[2] == "cl" else (True, False)
for cl in modes:
    t = run(cl)
    flop = (8.0 + 2.0) * B * 256 * 256 * 64 * 64 * 9 * 3
    print(f"{'channels-last kernels' if cl else 'planar (r01) kernels '}: {t * 1e3:8.3f} ms per step, {B / t:9.1f} images/s, "
          f"{flop / t / 1e12:6.1f} TFLOP/s (complex + variance convolutions)  [CplxConv2dVD(64,64,3) B={B} 256x256 bf16, LRT fwd + KL + bwd]")
