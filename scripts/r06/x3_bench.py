"""float32 layers: exact float32-MFMA kernels vs split operands on the bf16 pipe (x3).  python scripts/r06/x3_bench.py [log2B F]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from cplxmodule_amd import Cplx, fp32_mode  # noqa: E402
from cplxmodule_amd.nn import relevance as rel  # noqa: E402


def run(B, F, mode, steps=3, vd=True):
    torch.manual_seed(0)
    dev = "cuda"
    layer = (rel.CplxLinearVD(F, F) if vd else __import__("cplxmodule_amd").nn.CplxLinear(F, F)).to(dev)
    if vd:
        with torch.no_grad():
            layer.log_sigma2.uniform_(-12, 4)
    x = Cplx(torch.randn(B, F, device=dev).requires_grad_(True), torch.randn(B, F, device=dev).requires_grad_(True))
    klw = torch.tensor(1e-3, device=dev)

    def step():
        layer.zero_grad(set_to_none=True)
        x.real.grad = x.imag.grad = None
        with fp32_mode(mode):
            y = layer(x)
            if vd:
                kl = sum(rel.penalties(layer))
        gr, gi = y.real.detach() * 2, y.imag.detach() * 2
        if vd:
            torch.autograd.backward((y.real, y.imag, kl), (gr, gi, klw))
        else:
            torch.autograd.backward((y.real, y.imag), (gr, gi))
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 3 * ((8 + 2) if vd else 8) * float(B) * F * F
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"B={B} F={F} vd={vd} mode={mode}: {dt * 1e3:9.2f} ms/step  {flop / dt / 1e12:7.1f} TFLOP/s  peak {peak:.1f} GiB", flush=True)
    del layer, x
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    if len(sys.argv) > 2:
        shapes = [(1 << int(sys.argv[1]), int(sys.argv[2]))]
    else:
        shapes = [(8192, 4096), (1 << 17, 2048)]
    import os
    only = os.environ.get("X3_ONLY")          # e.g. X3_ONLY=x3 : the VD layer in that mode only (profiling)
    for B, F in shapes:
        for vd in ((True,) if only else (False, True)):
            for mode in ((only,) if only else ("exact", "x3")):
                run(B, F, mode, vd=vd)
