"""Batch-norm passes at BASELINE configs[2]'s shape (256 x 64 x 254 x 254 complex bf16, channels-last), one library build
per process (CPLXAMD_LIB=<lib.so>): forward = moment pass (reads x) + apply pass (reads x, writes y), backward = sums pass
(reads x, g) + apply pass (reads x, g, writes dx).  ms and GB/s of algorithmic bytes, HIP events, median of 9."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd.bn import CplxBatchNormFn  # noqa: E402

dev = "cuda"
B, F, H, W = (int(v) for v in os.environ.get("SHAPE", "256,64,254,254").split(","))
bf = torch.bfloat16
fmt = torch.channels_last
xr = torch.randn(B, F, H, W, device=dev).to(bf).contiguous(memory_format=fmt).requires_grad_(True)
xi = torch.randn(B, F, H, W, device=dev).to(bf).contiguous(memory_format=fmt).requires_grad_(True)
gr = torch.randn(B, F, H, W, device=dev).to(bf).contiguous(memory_format=fmt)
gi = torch.randn(B, F, H, W, device=dev).to(bf).contiguous(memory_format=fmt)
weight = torch.randn(2, 2, F, device=dev).requires_grad_(True)
bias = torch.randn(2, F, device=dev).requires_grad_(True)
rm, rv = torch.zeros(2, F, device=dev), torch.ones(2, 2, F, device=dev)
plane = B * F * H * W * 2          # bytes of one bf16 plane


def run():
    tf, tb = [], []
    for _ in range(11):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        yr, yi = CplxBatchNormFn.apply(xr, xi, weight, bias, rm, rv, True, 0.1, 1e-5)
        e[1].record()
        xr.grad = xi.grad = weight.grad = bias.grad = None
        e[2].record()
        torch.autograd.backward((yr, yi), (gr, gi))
        e[3].record()
        torch.cuda.synchronize()
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[2].elapsed_time(e[3]))
        del yr, yi
    med = lambda t: sorted(t[2:])[len(t[2:]) // 2]  # noqa: E731
    return med(tf), med(tb)


f, b = run()
print(f"{os.environ.get('CPLXAMD_LIB', 'libcplxamd.so'):40s} fwd {f:7.3f} ms {6 * plane / f / 1e6:7.0f} GB/s | bwd {b:7.3f} ms {10 * plane / b / 1e6:7.0f} GB/s")
