"""The persistent bf16 GEMM (csrc/gemm_bf16_persist.h) on shapes that take it (full tiles, more output tiles than
CUs) for all three ring phases R = (K / 32) % 3 and both instantiated layouts (complex: forward (N,N) and input
gradient (N,T) conj, bf16 out; real: forward (N,N) fp32 / bf16 out and input gradient (N,T) bf16 out), with and without
bias: against float64 numpy on sampled rows, and bit for bit against the one-tile-per-workgroup kernel -- both
run the same MFMA sequence per output element -- which is reached through an output with ldc = N + 4 (rows not
16-byte aligned: the persistent launcher declines those)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M, N = 8192, 4096           # 32 x 32 complex tiles (256 x 128), 32 x 16 real tiles (256 x 256): > 256 CUs


def _f(t):
    return t.double().cpu().numpy()


@pytest.mark.parametrize("K", [384, 416, 448, 800])          # 12, 13, 14, 25 K tiles of 32: R = 0, 1, 2, 1
@pytest.mark.parametrize("bias", [False, True])
def test_persistent_forward_layout(K, bias):
    from cplxmodule_amd import ops
    from cplxmodule_amd._lib import BF16, F32, call, ptr, stream_ptr
    dev, bf = "cuda", torch.bfloat16
    torch.manual_seed(K)
    ar, ai = (torch.randn(M, K, device=dev).to(bf) for _ in range(2))
    br, bi = (torch.randn(N, K, device=dev).mul(0.1).to(bf) for _ in range(2))
    b_r, b_i = torch.randn(N, device=dev), torch.randn(N, device=dev)
    rows = torch.randint(0, M, (16,), device=dev)
    R = rows.cpu().numpy()
    A, Bm = _f(ar) + 1j * _f(ai), _f(br) + 1j * _f(bi)
    yr, yi = ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, N, K, bias=(b_r, b_i) if bias else None, out_dtype=bf)
    ref = A[R] @ Bm.T + ((_f(b_r) + 1j * _f(b_i)) if bias else 0)
    got = _f(yr[rows]) + 1j * _f(yi[rows])
    assert np.abs(got - ref).max() <= 6e-3 * np.abs(ref).max()          # one bf16 rounding of the output
    pr, pi = (torch.zeros(M, N + 4, device=dev, dtype=bf) for _ in range(2))
    call("cplxamd_cgemm", ptr(ar), ptr(ai), K, 1, ptr(br), ptr(bi), K, 1, ptr(b_r) if bias else None,
         ptr(b_i) if bias else None, ptr(pr), ptr(pi), N + 4, M, N, K, 0, BF16, BF16, 0, 0, None, 0, stream_ptr())
    assert torch.equal(pr[:, :N], yr) and torch.equal(pi[:, :N], yi)
    assert not pr[:, N:].any() and not pi[:, N:].any()                  # nothing written past a row
    # real forward, fp32 out
    s2 = ops.rgemm(ar, (K, 1), br, (K, 1), M, N, K, bias=b_r if bias else None)
    ref = _f(ar)[R] @ _f(br).T + (_f(b_r) if bias else 0)
    assert np.abs(_f(s2[rows]) - ref).max() <= 2e-5 * np.abs(ref).max()
    ps = torch.zeros(M, N + 1, device=dev)
    call("cplxamd_rgemm", ptr(ar), K, 1, ptr(br), K, 1, ptr(b_r) if bias else None, None, ptr(ps), N + 1, M, N, K,
         BF16, F32, 0, None, 0, stream_ptr())
    assert torch.equal(ps[:, :N], s2) and not ps[:, N:].any()
    # real forward, bf16 out (mean GEMM of the real layers; variance GEMM of every bf16 LRT layer)
    mu = ops.rgemm(ar, (K, 1), br, (K, 1), M, N, K, bias=b_r if bias else None, out_dtype=bf)
    assert np.abs(_f(mu[rows]) - ref).max() <= 6e-3 * np.abs(ref).max()
    assert torch.equal(mu, s2.to(bf))                                   # same accumulators, one rounding
    pm = torch.zeros(M, N + 4, device=dev, dtype=bf)
    call("cplxamd_rgemm", ptr(ar), K, 1, ptr(br), K, 1, ptr(b_r) if bias else None, None, ptr(pm), N + 4, M, N, K,
         BF16, BF16, 0, None, 0, stream_ptr())
    assert torch.equal(pm[:, :N], mu) and not pm[:, N:].any()


@pytest.mark.parametrize("kt", [126, 127, 128])               # K tiles of the contraction over O: R = 0, 1, 2
def test_persistent_input_gradient_layout(kt):
    from cplxmodule_amd import ops
    from cplxmodule_amd._lib import BF16, call, ptr, stream_ptr
    dev, bf = "cuda", torch.bfloat16
    I, O = 4096, 32 * kt
    torch.manual_seed(kt)
    gr, gi = (torch.randn(M, O, device=dev).to(bf) for _ in range(2))
    wr, wi = (torch.randn(O, I, device=dev).mul(0.1).to(bf) for _ in range(2))
    rows = torch.randint(0, M, (16,), device=dev)
    R = rows.cpu().numpy()
    dr, di = ops.cgemm(gr, gi, (O, 1), wr, wi, (1, I), M, I, O, conj_b=True, out_dtype=bf)   # dX = G conj(W)
    ref = (_f(gr) + 1j * _f(gi))[R] @ (_f(wr) - 1j * _f(wi))
    got = _f(dr[rows]) + 1j * _f(di[rows])
    assert np.abs(got - ref).max() <= 6e-3 * np.abs(ref).max()
    pr, pi = (torch.zeros(M, I + 4, device=dev, dtype=bf) for _ in range(2))
    call("cplxamd_cgemm", ptr(gr), ptr(gi), O, 1, ptr(wr), ptr(wi), 1, I, None, None, ptr(pr), ptr(pi), I + 4,
         M, I, O, 1, BF16, BF16, 0, 0, None, 0, stream_ptr())
    assert torch.equal(pr[:, :I], dr) and torch.equal(pi[:, :I], di)
    dx = ops.rgemm(gr, (O, 1), wr, (1, I), M, I, O, out_dtype=bf)                            # ga = gs2 S
    ref = _f(gr)[R] @ _f(wr)
    assert np.abs(_f(dx[rows]) - ref).max() <= 6e-3 * np.abs(ref).max()
    px = torch.zeros(M, I + 4, device=dev, dtype=bf)
    call("cplxamd_rgemm", ptr(gr), O, 1, ptr(wr), 1, I, None, None, ptr(px), I + 4, M, I, O, BF16, BF16, 0,
         None, 0, stream_ptr())
    assert torch.equal(px[:, :I], dx) and not px[:, I:].any()


def test_persistent_switch_is_bit_identical_and_restores():
    """cplxamd_gemm_set_persistent(0) (what the data-parallel hook does while RCCL collectives are in flight) selects
    the one-workgroup-per-tile kernels: same results bit for bit; the call returns the previous setting."""
    from cplxmodule_amd import ops, _lib
    lib = _lib.load()
    dev, bf = "cuda", torch.bfloat16
    torch.manual_seed(5)
    K = 416
    ar, ai = (torch.randn(M, K, device=dev).to(bf) for _ in range(2))
    br, bi = (torch.randn(N, K, device=dev).mul(0.1).to(bf) for _ in range(2))
    on = ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, N, K, out_dtype=bf)
    assert lib.cplxamd_gemm_set_persistent(0) == 1
    try:
        off = ops.cgemm(ar, ai, (K, 1), br, bi, (K, 1), M, N, K, out_dtype=bf)
        assert lib.cplxamd_gemm_set_persistent(0) == 0
    finally:
        assert lib.cplxamd_gemm_set_persistent(1) == 0
    assert lib.cplxamd_gemm_set_persistent(1) == 1
    assert torch.equal(on[0], off[0]) and torch.equal(on[1], off[1])
