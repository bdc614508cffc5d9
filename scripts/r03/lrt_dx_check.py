"""cplxamd_cgemm_lrt_dx (dX = G conj(W) + 2 X ga in the persistent kernel's epilogue) vs cplxamd_cgemm + cplxamd_lrt_dx_accum:
bit-identical?  time per call (interleaved rounds, median)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import ops
dev, bf = "cuda", torch.bfloat16
B, I, O = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 4096, 4096)))
torch.manual_seed(0)
gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
wr, wi = (torch.randn(O, I, device=dev).mul(0.02).to(bf) for _ in range(2))
xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
ga = torch.randn(B, I, device=dev).mul(0.3).to(bf)

def two():
    dxr, dxi = ops._cplx_linear_dx(gr, gi, wr, wi, bf)
    ops.lrt_dx_accum(dxr, dxi, xr, xi, ga)
    return dxr, dxi

def one():
    return ops._cplx_lrt_dx(gr, gi, wr, wi, xr, xi, ga)

a, b = two(), one()
torch.cuda.synchronize()
print("bit-identical:", torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), " max |diff|", float((a[0].float() - b[0].float()).abs().max()))
def med(fn, n=8):
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[n // 2]
for _ in range(3):
    print(f"two kernels {med(two):.4f} ms   fused {med(one):.4f} ms")
