"""[CplxConv2d(64, 64, 3, padding 1) -> CplxBatchNorm2d -> CplxModReLU] x N on bf16 channels-last images: forward +
backward time per step, with the channels-last kernels and with the planar (round-1) path.
    python scripts/cl_chain_bench.py [B=32] [N=3] [H=W=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import Cplx, nn, conv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda"


def run(cl):
    conv._CL_ENABLED = cl
    torch.manual_seed(0)
    layers = []
    for _ in range(N):
        layers += [nn.CplxConv2d(64, 64, 3, padding=1), nn.CplxBatchNorm2d(64), nn.CplxModReLU(0.1)]
    net = torch.nn.Sequential(*layers).to(dev)
    fmt = torch.channels_last if cl else torch.contiguous_format
    mk = lambda: torch.randn(B, 64, HW, HW, device=dev).bfloat16().contiguous(memory_format=fmt)  # noqa: E731
    x = Cplx(mk().requires_grad_(True), mk().requires_grad_(True))

    def step():
        net.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        y = net(x)
        torch.autograd.backward((y.real, y.imag), (y.real.detach(), y.imag.detach()))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 8


for cl in (True, False):
    t = run(cl)
    flop = 8.0 * B * HW * HW * 64 * 64 * 9 * 3 * N
    print(f"{'channels-last kernels' if cl else 'planar (r01) kernels '}: {t * 1e3:8.3f} ms per step, {B / t:9.1f} images/s, "
          f"{flop / t / 1e12:6.1f} TFLOP/s of convolution work incl. batch-norm and activations  [B={B}, {N} blocks, {HW}x{HW}]")
