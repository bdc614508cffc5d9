"""Round 4: the one-wave-per-SIMD GEMM family (gemm_bf16_w4.hip) against the 8-wave family, same library, same process.

1. bit-exactness: every launch of the bench step (and a few other shapes) with cplxamd_gemm_set_family(0) and (1) must
   give identical bits (same MFMA sequence per accumulator);
2. interleaved timing (cdna_hip_programming.md rule 24): ROUNDS x PER launches per family and shape, median / min.

    python scripts/r04/w4_ab.py [lib.so]         (env: ROUNDS, PER, ONLY=comma list, SHAPES=bench|cfg4|all)
"""
import ctypes
import os
import statistics
import sys
from ctypes import c_int, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cplxmodule_amd import _lib as L  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "7"))
PER = int(os.environ.get("PER", "8"))
ONLY = os.environ.get("ONLY", "")
SHAPES = os.environ.get("SHAPES", "bench")


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("cplxamd_cgemm", "cplxamd_rgemm", "cplxamd_cgemm_ex", "cplxamd_rgemm_ex", "cplxamd_cgemm_lrt_dx",
                 "cplxamd_rgemm_lrt_dx", "cplxamd_gemm_set_family", "cplxamd_gemm_set_persistent"):
        fn = getattr(lib, name)
        fn.argtypes = L.SIGNATURES[name]
        fn.restype = c_int
    return lib


def make(lib, B, I, O, dev="cuda"):
    torch.manual_seed(0)
    bf = torch.bfloat16
    bound = (1.0 / (2 * I)) ** 0.5
    xr, xi = (torch.randn(B, I, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.empty(O, I, device=dev).uniform_(-bound, bound).to(bf) for _ in range(2))
    gr, gi = (torch.randn(B, O, device=dev).to(bf) for _ in range(2))
    br, bi = (torch.randn(O, device=dev) for _ in range(2))
    a2 = (xr.float() ** 2 + xi.float() ** 2).to(bf)
    S = torch.empty(O, I, device=dev).uniform_(-12, 4).exp().to(bf)
    gs2 = torch.randn(B, O, device=dev).to(bf)
    ga = torch.randn(B, I, device=dev).to(bf)
    ls2 = torch.empty(O, I, device=dev).uniform_(-12, 4)
    kl0 = [torch.randn(O, I, device=dev) for _ in range(2)]
    y_bf = [torch.empty(B, O, device=dev, dtype=bf) for _ in range(2)]
    dx_bf = [torch.empty(B, I, device=dev, dtype=bf) for _ in range(2)]
    dw_f = [torch.empty(O, I, device=dev) for _ in range(2)]
    s2_bf = torch.empty(B, O, device=dev, dtype=bf)
    beta = torch.tensor(1e-3, device=dev)
    p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
    st = c_void_p(torch.cuda.current_stream().cuda_stream)

    def chk(rc):
        assert rc == 0, rc

    def c_fwd():
        chk(lib.cplxamd_cgemm(p(xr), p(xi), I, 1, p(wr), p(wi), I, 1, p(br), p(bi), p(y_bf[0]), p(y_bf[1]), O, B, O, I, 0,
                              L.BF16, L.BF16, 0, 0, None, 0, st))
        return y_bf

    def c_dgrad():
        chk(lib.cplxamd_cgemm(p(gr), p(gi), O, 1, p(wr), p(wi), 1, I, None, None, p(dx_bf[0]), p(dx_bf[1]), I, B, I, O, 1,
                              L.BF16, L.BF16, 0, 0, None, 0, st))
        return dx_bf

    def c_dgrad_lrt():
        chk(lib.cplxamd_cgemm_lrt_dx(p(gr), p(gi), O, 1, p(wr), p(wi), 1, I, p(xr), p(xi), p(ga), I, p(dx_bf[0]), p(dx_bf[1]),
                                     I, B, I, O, L.BF16, st))
        return dx_bf

    def c_wgrad_kl():      # dW = G^T conj(X) + beta * dW_kl
        dw_f[0].copy_(kl0[0]); dw_f[1].copy_(kl0[1])
        chk(lib.cplxamd_cgemm_ex(p(gr), p(gi), 1, O, p(xr), p(xi), 1, I, None, None, None, p(dw_f[0]), p(dw_f[1]), I,
                                 O, I, B, 1, L.BF16, L.F32, 1, p(beta), 0, None, 0, st))
        return dw_f

    def r_fwd():
        chk(lib.cplxamd_rgemm(p(a2), I, 1, p(S), I, 1, None, None, p(s2_bf), O, B, O, I, L.BF16, L.BF16, 0, None, 0, st))
        return [s2_bf]

    def r_dgrad():
        chk(lib.cplxamd_rgemm(p(gs2), O, 1, p(S), 1, I, None, None, p(dx_bf[0]), I, B, I, O, L.BF16, L.BF16, 0, None, 0, st))
        return dx_bf[:1]

    def r_wgrad_kl():      # dls2 = (gs2^T |x|^2) * exp(ls2) + beta * dls2_kl
        dw_f[0].copy_(kl0[0])
        chk(lib.cplxamd_rgemm_ex(p(gs2), 1, O, p(a2), 1, I, None, p(ls2), 1, p(dw_f[0]), I, O, I, B, L.BF16, L.F32,
                                 1, p(beta), None, 0, st))
        return dw_f[:1]

    lib.cplxamd_gemm_ws_bytes.argtypes = L.SIGNATURES["cplxamd_gemm_ws_bytes"]
    lib.cplxamd_gemm_ws_bytes.restype = ctypes.c_int64
    nws = max(int(lib.cplxamd_gemm_ws_bytes(O, I, B, 1, L.BF16, L.F32)), int(lib.cplxamd_gemm_ws_bytes(O, I, B, 0, L.BF16, L.F32)), 16)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)

    def c_wgrad_ws():      # the weight gradient as the layers launch it: with the split-K workspace (used when few tiles + long K)
        dw_f[0].copy_(kl0[0]); dw_f[1].copy_(kl0[1])
        chk(lib.cplxamd_cgemm_ex(p(gr), p(gi), 1, O, p(xr), p(xi), 1, I, None, None, None, p(dw_f[0]), p(dw_f[1]), I,
                                 O, I, B, 1, L.BF16, L.F32, 1, p(beta), 0, p(ws), nws, st))
        return dw_f

    def r_wgrad_ws():
        dw_f[0].copy_(kl0[0])
        chk(lib.cplxamd_rgemm_ex(p(gs2), 1, O, p(a2), 1, I, None, p(ls2), 1, p(dw_f[0]), I, O, I, B, L.BF16, L.F32,
                                 1, p(beta), p(ws), nws, st))
        return dw_f[:1]

    extra = {"c_wgrad_ws": (c_wgrad_ws, 8.0), "r_wgrad_ws": (r_wgrad_ws, 2.0)} if os.environ.get("WS") else {}
    return {**extra, "c_fwd": (c_fwd, 8.0), "c_dgrad": (c_dgrad, 8.0), "c_dgrad_lrt": (c_dgrad_lrt, 8.0), "c_wgrad_kl": (c_wgrad_kl, 8.0),
            "r_fwd": (r_fwd, 2.0), "r_dgrad": (r_dgrad, 2.0), "r_wgrad_kl": (r_wgrad_kl, 2.0)}


def run(lib, tag, B, I, O, time_it=True):
    shapes = make(lib, B, I, O)
    if ONLY:
        shapes = {k: v for k, v in shapes.items() if k in ONLY.split(",")}
    print(f"## {tag}: B={B} I={I} O={O}", flush=True)
    # ---- bit-exactness
    bad = 0
    for s, (fn, _) in shapes.items():
        lib.cplxamd_gemm_set_family(0)
        ref = [t.clone() for t in fn()]
        lib.cplxamd_gemm_set_family(-1)
        out = [t.clone() for t in fn()]
        lib.cplxamd_gemm_set_persistent(0)
        out1 = [t.clone() for t in fn()]
        lib.cplxamd_gemm_set_persistent(1)
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(out, out1)):
            bad += 1
            print(f"   {s:12s} MISMATCH between the persistent and the one-workgroup-per-tile launch", flush=True)
        same = all(torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32),
                               b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)) for a, b in zip(ref, out))
        fin = all(torch.isfinite(b.float()).all().item() for b in out)
        if not same:
            bad += 1
            d = max((a.float() - b.float()).abs().max().item() for a, b in zip(ref, out))
            n = sum((a != b).sum().item() for a, b in zip(ref, out))
            print(f"   {s:12s} MISMATCH: {n} elements differ, max abs diff {d:.4g}, finite={fin}", flush=True)
        else:
            print(f"   {s:12s} identical bits (finite={fin})", flush=True)
    if not time_it:
        return bad
    # ---- timing
    # w8: the 8-wave kernels (persistent where they have that form); w4: one wave per SIMD, persistent; w4o: one workgroup per tile
    fams = [("w8", 0), ("w4", 1), ("w4o", 2)] if os.environ.get("W4O") else [("w8", 0), ("w4", 1)]
    times = {(n, s): [] for n, _ in fams for s in shapes}
    for n, f in fams:
        lib.cplxamd_gemm_set_family(-1 if f else 0)
        lib.cplxamd_gemm_set_persistent(0 if f == 2 else 1)
        for s, (fn, _) in shapes.items():
            for _ in range(3):
                fn()
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for s, (fn, _) in shapes.items():
            order = fams if r % 2 == 0 else fams[::-1]
            for n, f in order:
                lib.cplxamd_gemm_set_family(-1 if f else 0)
                lib.cplxamd_gemm_set_persistent(0 if f == 2 else 1)
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(PER):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[(n, s)].append(e0.elapsed_time(e1) / PER)
    flop = B * I * O
    lib.cplxamd_gemm_set_persistent(1)
    print(f"# {ROUNDS} interleaved rounds x {PER} launches; median ms (min ms) [TF/s at the median; 8MNK complex, 2MNK real]")
    print("family".ljust(8) + "".join(s.rjust(30) for s in shapes))
    for n, _ in fams:
        row = n.ljust(8)
        for s, (_, mult) in shapes.items():
            med, mn = statistics.median(times[(n, s)]), min(times[(n, s)])
            row += f"{med:.4f} ({mn:.4f}) [{mult * flop / med / 1e9:6.0f}]".rjust(30)
        print(row, flush=True)
    return bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(L.__file__), "libcplxamd.so")
    lib = load(path)
    bad = 0
    if SHAPES in ("bench", "all"):
        bad += run(lib, "bench step (configs[1])", 8192, 4096, 4096)
    if SHAPES in ("small", "all"):
        # K tile counts 4 (tail only), 12, 14 (6 + 6 + 2), 16, 20: every entry / exit of the 6-tile ring loop
        bad += run(lib, "small: K = 384 / 512", 512, 384, 512, time_it=False)
        bad += run(lib, "small: one tile, K = 128", 256, 128, 256, time_it=False)
        bad += run(lib, "small: K = 448 / 512 / 768", 768, 448, 512, time_it=False)
        bad += run(lib, "small: K = 640 / 256 / 1024", 1024, 640, 256, time_it=False)
    if SHAPES in ("cfg4", "all"):
        bad += run(lib, "configs[3] at batch 2^16", 65536, 2048, 2048)
    print("RESULT", "ok" if bad == 0 else f"{bad} mismatching launches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
