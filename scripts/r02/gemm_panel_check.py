"""K-panel-major operands for the bf16 GEMMs: results equal the row-major launch bit for bit (same MFMA sequence), timing
of the complex / real forward and input-gradient layouts with none / B / A / both operands in panel layout."""
import ctypes, os, sys
from ctypes import c_int, c_int64, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cplxmodule_amd import _lib as L, ops

lib = L.load()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
M, N, K = 8192, 4096, 4096
p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
st = c_void_p(torch.cuda.current_stream().cuda_stream)
panel = lambda t: t.view(t.shape[0], t.shape[1] // 32, 32).permute(1, 0, 2).contiguous()  # noqa: E731


def timeit(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            f()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n)
    return sorted(ts)[2]


ar, ai = (torch.randn(M, K, device=dev).to(bf) for _ in range(2))
br, bi = ((torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(2))
apr, api, bpr, bpi = panel(ar), panel(ai), panel(br), panel(bi)
yr, yi = (torch.empty(M, N, device=dev, dtype=bf) for _ in range(2))
ref = ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, N, K, out_dtype=bf)
for name, (a0, a1, ap), (b0, b1, bp) in (("none", (ar, ai, 0), (br, bi, 0)), ("B", (ar, ai, 0), (bpr, bpi, N)),
                                         ("A", (apr, api, M), (br, bi, 0)), ("A+B", (apr, api, M), (bpr, bpi, N))):
    def f():
        rc = lib.cplxamd_cgemm_panel(p(a0), p(a1), K, 1, ap, p(b0), p(b1), K, 1, bp, None, None, p(yr), p(yi), N, M, N, K,
                                     0, L.BF16, L.BF16, st)
        assert rc == 0, rc
    f()
    ok = torch.equal(yr, ref[0]) and torch.equal(yi, ref[1])
    t = timeit(f)
    print(f"complex forward (N,N) panel {name:5s}: {t:.4f} ms  {8.0 * M * N * K / t / 1e9:6.0f} TF/s  frac {8.0 * M * N * K / t / 1e9 / 2500:.3f}  bit-identical {ok}")
# input gradient layout (N,T conj): A = G [M, K] (panel-able), B = W read K-major (already whole lines)
wr, wi = ((torch.randn(K, N, device=dev) * 0.02).to(bf) for _ in range(2))       # [O = K][I = N]
ref = ops.cgemm(ar, ai, (K, 1), wr, wi, (1, N), M, N, K, conj_b=True, out_dtype=bf)
for name, (a0, a1, ap) in (("none", (ar, ai, 0)), ("A", (apr, api, M))):
    def f():
        rc = lib.cplxamd_cgemm_panel(p(a0), p(a1), K, 1, ap, p(wr), p(wi), 1, N, 0, None, None, p(yr), p(yi), N, M, N, K,
                                     1, L.BF16, L.BF16, st)
        assert rc == 0, rc
    f()
    ok = torch.equal(yr, ref[0]) and torch.equal(yi, ref[1])
    t = timeit(f)
    print(f"complex input gradient (N,T) panel {name:5s}: {t:.4f} ms  {8.0 * M * N * K / t / 1e9:6.0f} TF/s  bit-identical {ok}")
# real (variance) forward
y = torch.empty(M, N, device=dev, dtype=bf)
ref = ops.rgemm(ar, (K, 1), br, (K, 1), M, N, K, out_dtype=bf)
for name, (a0, ap), (b0, bp) in (("none", (ar, 0), (br, 0)), ("B", (ar, 0), (bpr, N)), ("A", (apr, M), (br, 0)), ("A+B", (apr, M), (bpr, N))):
    def f():
        rc = lib.cplxamd_rgemm_panel(p(a0), K, 1, ap, p(b0), K, 1, bp, None, p(y), N, M, N, K, L.BF16, L.BF16, st)
        assert rc == 0, rc
    f()
    ok = torch.equal(y, ref)
    t = timeit(f)
    print(f"real forward (N,N) panel {name:5s}: {t:.4f} ms  {2.0 * M * N * K / t / 1e9:6.0f} TF/s  frac {2.0 * M * N * K / t / 1e9 / 2500:.3f}  bit-identical {ok}")
