"""Correctness of the bf16 GEMM launches of one build of the library (ctypes; A/B builds) against float64 numpy on
sampled rows: complex forward (bias) / dgrad (conj, K-major B) / wgrad (K-major both), real forward / dgrad, at the
bench shape and at a shape with odd tile counts.  usage: gemm_check.py <lib.so>"""
import ctypes, os, sys
from ctypes import c_int, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cplxmodule_amd import _lib as L

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for n in ("cplxamd_cgemm", "cplxamd_rgemm"):
    getattr(lib, n).argtypes = L.SIGNATURES[n]; getattr(lib, n).restype = c_int
p = lambda t: c_void_p(t.data_ptr()) if t is not None else None
st = c_void_p(torch.cuda.current_stream().cuda_stream)
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(1)
worst = 0.0
for (B, I, O) in ((8192, 4096, 4096), (2304, 1536, 1280), (512, 256, 384)):
    xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.randn(O, I, device=dev).mul(0.05).to(bf) for _ in range(2))
    gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
    br, bi = torch.randn(O, device=dev), torch.randn(O, device=dev)
    rows = torch.randint(0, B, (24,), device=dev)
    f = lambda t: t.double().cpu().numpy()
    X, W, G = f(xr) + 1j * f(xi), f(wr) + 1j * f(wi), f(gr) + 1j * f(gi)
    R = rows.cpu().numpy()
    # complex forward, bf16 out, bias
    yr, yi = torch.empty(B, O, device=dev, dtype=bf), torch.empty(B, O, device=dev, dtype=bf)
    assert lib.cplxamd_cgemm(p(xr), p(xi), I, 1, p(wr), p(wi), I, 1, p(br), p(bi), p(yr), p(yi), O, B, O, I, 0, L.BF16, L.BF16, 0, 0, None, 0, st) == 0
    ref = X[R] @ W.T + (f(br) + 1j * f(bi))
    got = f(yr[rows]) + 1j * f(yi[rows])
    e1 = np.abs(got - ref).max() / np.abs(ref).max()
    # complex dgrad: dX = G conj(W), W read K-major, bf16 out
    dr, di = torch.empty(B, I, device=dev, dtype=bf), torch.empty(B, I, device=dev, dtype=bf)
    assert lib.cplxamd_cgemm(p(gr), p(gi), O, 1, p(wr), p(wi), 1, I, None, None, p(dr), p(di), I, B, I, O, 1, L.BF16, L.BF16, 0, 0, None, 0, st) == 0
    ref = G[R] @ W.conj()
    e2 = np.abs(f(dr[rows]) + 1j * f(di[rows]) - ref).max() / np.abs(ref).max()
    # complex wgrad: dW = G^T conj(X), f32 out
    cr, ci = torch.empty(O, I, device=dev), torch.empty(O, I, device=dev)
    assert lib.cplxamd_cgemm(p(gr), p(gi), 1, O, p(xr), p(xi), 1, I, None, None, p(cr), p(ci), I, O, I, B, 1, L.BF16, L.F32, 0, 0, None, 0, st) == 0
    ro = torch.randint(0, O, (16,), device=dev); RO = ro.cpu().numpy()
    ref = G[:, RO].T @ X.conj()
    e3 = np.abs(f(cr[ro]) + 1j * f(ci[ro]) - ref).max() / np.abs(ref).max()
    # real forward f32 out / dgrad bf16 out
    s2 = torch.empty(B, O, device=dev)
    assert lib.cplxamd_rgemm(p(xr), I, 1, p(wr), I, 1, None, None, p(s2), O, B, O, I, L.BF16, L.F32, 0, None, 0, st) == 0
    ref = f(xr)[R] @ f(wr).T
    e4 = np.abs(f(s2[rows]) - ref).max() / np.abs(ref).max()
    dx = torch.empty(B, I, device=dev, dtype=bf)
    assert lib.cplxamd_rgemm(p(gr), O, 1, p(wr), 1, I, p(torch.zeros(I, device=dev)), None, p(dx), I, B, I, O, L.BF16, L.BF16, 0, None, 0, st) == 0
    ref = f(gr)[R] @ f(wr)
    e5 = np.abs(f(dx[rows]) - ref).max() / np.abs(ref).max()
    torch.cuda.synchronize()
    print(f"{os.path.basename(sys.argv[1])} B={B} I={I} O={O}: cfwd {e1:.2e} cdgrad {e2:.2e} cwgrad {e3:.2e} rfwd {e4:.2e} rdgrad {e5:.2e}")
    worst = max(worst, e1, e2, e3, e4, e5)
assert worst < 1.5e-2, worst      # bf16 outputs: half an ulp of bf16 relative to the largest entry
print("gemm_check OK")
