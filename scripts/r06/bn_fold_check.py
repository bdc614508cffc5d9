"""The batch-norm backward folded into the convolution's weight gradient (csrc/conv_cl_wgrad.hip FOLD) against the three
separate launches: values on a few geometries, then the cfg3 step both ways."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cplxmodule_amd import Cplx, nn, conv as cv  # noqa: E402

dev = "cuda"


def run(B, Ci, Co, H, W, pad, fold, seed=0, train=True):
    cv._BN_FOLD = fold
    torch.manual_seed(seed)
    layer, bn = nn.CplxConv2d(Ci, Co, 3, padding=pad).to(dev), nn.CplxBatchNorm2d(Co).to(dev)
    with torch.no_grad():
        bn.weight.add_(0.3 * torch.randn_like(bn.weight)); bn.bias.add_(0.3 * torch.randn_like(bn.bias))
    mk = lambda: (torch.randn(B, Ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
                  .requires_grad_(True))
    x = Cplx(mk(), mk())
    if not train:
        bn(layer(x)); bn.eval()
    y = bn(layer(x))
    g = (torch.randn_like(y.real), torch.randn_like(y.imag))
    torch.autograd.backward((y.real, y.imag), g)
    return [x.real.grad, x.imag.grad, layer.weight.real.grad, layer.weight.imag.grad, layer.bias.real.grad,
            layer.bias.imag.grad, bn.weight.grad, bn.bias.grad]


cv._CL_FORCE = True
names = ["dx_r", "dx_i", "dW_r", "dW_i", "db_r", "db_i", "bn_dw", "bn_db"]
for shape in [(2, 64, 64, 64, 64, 0, True), (3, 64, 64, 40, 72, 1, True), (2, 128, 64, 66, 50, 1, True),
              (2, 64, 128, 48, 64, 0, True), (3, 128, 128, 64, 32, 1, False), (1, 64, 64, 130, 97, 0, True),
              (2, 64, 64, 64, 64, 1, False)]:
    a = run(*shape[:6], fold=False, train=shape[6])
    b = run(*shape[:6], fold=True, train=shape[6])
    line = []
    for n, u, v in zip(names, a, b):
        err = float((u.float() - v.float()).abs().max()) / max(float(u.float().abs().max()), 1e-30)
        line.append(f"{n} {'=' if torch.equal(u, v) else f'{err:.1e}'}")
    print(shape, " ".join(line), flush=True)

cv._CL_FORCE = False
for fold in (False, True, False, True):
    cv._BN_FOLD = fold
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
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"cfg3 bf16 batch 256, fold={fold}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms / step", flush=True)
    del x, layer, bn
    torch.cuda.empty_cache()
