"""Within-process interleaved A/B of several builds of libcplxamd.so on the bf16 GEMM launches of the
bench step (cdna_hip_programming.md rule 24: N variants x M rounds in ONE process, median and min).

    python scripts/gemm_ab.py base=cplxmodule_amd/libcplxamd.so u3=cplxmodule_amd/libcplxamd_u3.so ...

Shapes = what one bench.py step launches (B = 8192, I = O = 4096): complex forward (N,N) bf16 out
with bias, complex dgrad (N,T) conj bf16 out, complex wgrad (T,T) conj fp32 out, and the three real
(variance) GEMMs.  Operands as in the bench: x ~ N(0,1), W ~ U(+-sqrt(1/2I)), G ~ N(0,1)."""
import ctypes
import os
import statistics
import sys
from ctypes import c_int, c_int64, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

B, I, O = 8192, 4096, 4096
ROUNDS = int(os.environ.get("ROUNDS", "7"))
PER = int(os.environ.get("PER", "8"))
ONLY = os.environ.get("ONLY", "")          # comma list of shape names


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("cplxamd_cgemm", "cplxamd_rgemm", "cplxamd_cgemm_ex", "cplxamd_rgemm_ex"):
        fn = getattr(lib, name)
        fn.argtypes = L.SIGNATURES[name]
        fn.restype = c_int
    return lib


def main():
    libs = [(a.split("=")[0], load(a.split("=")[1])) for a in sys.argv[1:]]
    dev = "cuda"
    torch.manual_seed(0)
    bf = torch.bfloat16
    bound = (1.0 / (2 * I)) ** 0.5
    xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.empty(O, I, device=dev).uniform_(-bound, bound).to(bf) for _ in range(2))
    gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
    br, bi = (torch.zeros(O, device=dev) for _ in range(2))
    a2 = (xr.float() ** 2 + xi.float() ** 2).to(bf)
    S = torch.empty(O, I, device=dev).uniform_(-12, 4).exp().to(bf)
    gs2 = torch.randn(B, O, device=dev).to(bf)
    emul = torch.rand(O, I, device=dev)
    y_bf = [torch.empty(B, O, device=dev, dtype=bf) for _ in range(2)]
    dx_bf = [torch.empty(B, I, device=dev, dtype=bf) for _ in range(2)]
    dw_f = [torch.empty(O, I, device=dev) for _ in range(2)]
    s2_f = torch.empty(B, O, device=dev)
    p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
    st = c_void_p(torch.cuda.current_stream().cuda_stream)

    def cg(lib, ar, ai, ast, b_r, b_i, bst, bias, out, M, N, K, conj, odt):
        rc = lib.cplxamd_cgemm(p(ar), p(ai), ast[0], ast[1], p(b_r), p(b_i), bst[0], bst[1],
                               p(bias[0]) if bias else None, p(bias[1]) if bias else None, p(out[0]), p(out[1]),
                               N, M, N, K, int(conj), L.BF16, odt, 0, 0, None, 0, st)
        assert rc == 0, rc

    def rg(lib, a, ast, b, bst, em, out, M, N, K, odt):
        rc = lib.cplxamd_rgemm(p(a), ast[0], ast[1], p(b), bst[0], bst[1], None, p(em) if em is not None else None,
                               p(out), N, M, N, K, L.BF16, odt, 0, None, 0, st)
        assert rc == 0, rc

    beta = torch.tensor(1e-3, device=dev)
    ls2 = torch.empty(O, I, device=dev).uniform_(-12, 4)

    def cg_kl(lib):      # the weight gradient as the fused-KL layer launches it: dW = G^T conj(X) + beta * dW_kl
        rc = lib.cplxamd_cgemm_ex(p(gr), p(gi), 1, O, p(xr), p(xi), 1, I, None, None, None, p(dw_f[0]), p(dw_f[1]), I,
                                  O, I, B, 1, L.BF16, L.F32, 1, p(beta), 0, None, 0, st)
        assert rc == 0, rc

    def rg_kl(lib):      # dls2 = (gs2^T |x|^2) * exp(ls2) + beta * dls2_kl
        rc = lib.cplxamd_rgemm_ex(p(gs2), 1, O, p(a2), 1, I, None, p(ls2), 1, p(dw_f[0]), I, O, I, B, L.BF16, L.F32,
                                  1, p(beta), None, 0, st)
        assert rc == 0, rc

    shapes = {
        "c_fwd": (lambda lib: cg(lib, xr, xi, (I, 1), wr, wi, (I, 1), (br, bi), y_bf, B, O, I, False, L.BF16), 8.0),
        "c_dgrad": (lambda lib: cg(lib, gr, gi, (O, 1), wr, wi, (1, I), None, dx_bf, B, I, O, True, L.BF16), 8.0),
        "c_wgrad": (lambda lib: cg(lib, gr, gi, (1, O), xr, xi, (1, I), None, dw_f, O, I, B, True, L.F32), 8.0),
        "r_fwd": (lambda lib: rg(lib, a2, (I, 1), S, (I, 1), None, s2_f, B, O, I, L.F32), 2.0),
        "r_dgrad": (lambda lib: rg(lib, gs2, (O, 1), S, (1, I), None, dx_bf[0], B, I, O, L.BF16), 2.0),
        "r_wgrad": (lambda lib: rg(lib, gs2, (1, O), a2, (1, I), emul, dw_f[0], O, I, B, L.F32), 2.0),
        "c_wgrad_kl": (cg_kl, 8.0),
        "r_wgrad_kl": (rg_kl, 2.0),
    }
    if ONLY:
        shapes = {k: v for k, v in shapes.items() if k in ONLY.split(",")}
    times = {(n, s): [] for n, _ in libs for s in shapes}
    for n, lib in libs:                      # warm every build on every shape
        for s, (fn, _) in shapes.items():
            for _ in range(3):
                fn(lib)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for s, (fn, _) in shapes.items():
            order = libs if r % 2 == 0 else libs[::-1]
            for n, lib in order:
                fn(lib)                       # one untimed launch after the switch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(PER):
                    fn(lib)
                e1.record()
                torch.cuda.synchronize()
                times[(n, s)].append(e0.elapsed_time(e1) / PER)
    flop = B * I * O
    print(f"# {ROUNDS} interleaved rounds x {PER} launches; median ms (min ms) [TF/s at the median; 8MNK complex, 2MNK real]")
    print("variant".ljust(12) + "".join(s.rjust(30) for s in shapes) + "   sum3c".rjust(10))
    for n, _ in libs:
        row, tot = n.ljust(12), 0.0
        for s, (_, mult) in shapes.items():
            med, mn = statistics.median(times[(n, s)]), min(times[(n, s)])
            row += f"{med:.4f} ({mn:.4f}) [{mult * flop / med / 1e9:6.0f}]".rjust(30)
            if s in ("c_fwd", "c_dgrad", "c_wgrad"):
                tot += med
        print(row + f"{tot:10.4f}")


if __name__ == "__main__":
    main()
