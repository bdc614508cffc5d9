"""Per-kernel timings on one MI355X (HIP events on the current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cplxmodule_amd import ops


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = "cuda"
    torch.manual_seed(0)
    print("== complex bf16 GEMM (4M) ==")
    for (M, N, K) in [(8192, 4096, 4096), (4096, 4096, 8192), (8192, 8192, 8192), (2048, 2048, 2048)]:
        a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
        b = [torch.randn(N, K, device=dev).bfloat16() for _ in range(2)]
        out = (torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        t = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K, out=out))
        print(f"cgemm bf16 {M}x{N}x{K}: {t*1e3:.3f} ms  {8*M*N*K/t/1e12:.1f} TF/s")
    print("== real bf16 GEMM ==")
    M, N, K = 8192, 4096, 4096
    a, b = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16()
    t = timeit(lambda: ops.rgemm(a, (K, 1), b, (K, 1), M, N, K))
    print(f"rgemm bf16 {M}x{N}x{K}: {t*1e3:.3f} ms  {2*M*N*K/t/1e12:.1f} TF/s")
    print("== complex fp32 GEMM (generic) ==")
    M, N, K = 4096, 2048, 2048
    a = [torch.randn(M, K, device=dev) for _ in range(2)]
    b = [torch.randn(N, K, device=dev) for _ in range(2)]
    t = timeit(lambda: ops.cgemm(a[0], a[1], (K, 1), b[0], b[1], (K, 1), M, N, K), iters=5, warm=2)
    print(f"cgemm f32 {M}x{N}x{K}: {t*1e3:.3f} ms  {8*M*N*K/t/1e12:.1f} TF/s")
    print("== KL ==")
    for n in (2048, 4096, 8192, 16384):
        wr, wi = torch.randn(n, n, device=dev) * 0.05, torch.randn(n, n, device=dev) * 0.05
        ls2 = torch.empty(n, n, device=dev).uniform_(-12, 4)
        for kind in ("cplx_vd", "cplx_ard", "real_vd"):
            wi_ = wi if kind.startswith("cplx") else None
            nb = 12 if wi_ is not None else 8
            t = timeit(lambda: ops.kl_fwd(kind, wr, wi_, ls2))
            gs = torch.ones((), device=dev)
            t2 = timeit(lambda: ops.kl_bwd(kind, wr, wi_, ls2, g_scalar=gs))
            t3 = timeit(lambda: ops.kl_fwd_bwd(kind, wr, wi_, ls2))
            print(f"kl {kind} {n}^2: fwd {t*1e6:.1f} us {nb*n*n/t/1e9:.0f} GB/s | bwd {t2*1e6:.1f} us "
                  f"{2*nb*n*n/t2/1e9:.0f} GB/s | fused {t3*1e6:.1f} us {2*nb*n*n/t3/1e9:.0f} GB/s")
        t = timeit(lambda: ops.relevance_mask(wr, wi, ls2, 1.0))
        print(f"mask {n}^2: {t*1e6:.1f} us {16*n*n/t/1e9:.0f} GB/s")
    del wr, wi, ls2
    torch.cuda.empty_cache()
    print("== reparam ==")
    for (B, O) in [(8192, 4096), (1 << 17, 2048), (1 << 20, 2048)]:
        n = B * O
        for dt, nb_f, nb_b in ((torch.float32, 20, 16), (torch.bfloat16, 12, 10)):
            mu_r, mu_i = torch.randn(n, device=dev, dtype=dt), torch.randn(n, device=dev, dtype=dt)
            s2 = torch.rand(n, device=dev)
            t = timeit(lambda: ops.reparam_fwd(mu_r, mu_i, s2, None, 1, 1, inplace=True), iters=10, warm=3)
            t2 = timeit(lambda: ops.reparam_bwd(mu_r, mu_i, s2, None, 1, 1), iters=10, warm=3)
            print(f"reparam {dt} B={B} O={O}: fwd {t*1e3:.3f} ms {nb_f*n/t/1e9:.0f} GB/s | "
                  f"bwd {t2*1e3:.3f} ms {nb_b*n/t2/1e9:.0f} GB/s")
            del mu_r, mu_i, s2
            torch.cuda.empty_cache()
    print("== aux ==")
    x = torch.randn(8192, 4096, device=dev).bfloat16()
    t = timeit(lambda: ops.transpose2d(x))
    print(f"transpose bf16 8192x4096: {t*1e6:.1f} us {2*x.numel()*2/t/1e9:.0f} GB/s")
    w = torch.randn(4096, 4096, device=dev)
    t = timeit(lambda: ops.cast(w, torch.bfloat16))
    print(f"cast f32->bf16 4096^2: {t*1e6:.1f} us {6*w.numel()/t/1e9:.0f} GB/s")
    t = timeit(lambda: ops.colsum(x))
    print(f"colsum bf16 8192x4096: {t*1e6:.1f} us {x.numel()*2/t/1e9:.0f} GB/s")
    print("== bilinear reduction (T = [B, O, I1] complex) ==")
    for dt, nb in ((torch.float32, 4), (torch.bfloat16, 2)):
        for (B, O, I1) in [(8192, 256, 64), (65536, 64, 16), (1024, 256, 512)]:
            u = [torch.randn(B, I1, device=dev).to(dt) for _ in range(2)]
            tt = [torch.randn(B, O * I1, device=dev).to(dt) for _ in range(2)]
            g = [torch.randn(B, O, device=dev).to(dt) for _ in range(2)]
            tf = timeit(lambda: ops.bilinear_reduce_fwd(u, tt, None, B, O, I1, True))
            tb = timeit(lambda: ops.bilinear_reduce_bwd(u, tt, g, B, O, I1, True))
            n = B * O * I1
            print(f"bilinear_reduce {dt} B={B} O={O} I1={I1}: fwd {tf*1e6:.1f} us {2*nb*n/tf/1e9:.0f} GB/s | "
                  f"bwd {tb*1e6:.1f} us {4*nb*n/tb/1e9:.0f} GB/s")


if __name__ == "__main__":
    main()
