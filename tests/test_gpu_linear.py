"""Complex / real GEMM kernels and the linear layers against golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import cplxmodule_amd
    return cplxmodule_amd


def _scale_tol(ref, rtol):
    return dict(rtol=rtol, atol=rtol * float(np.abs(ref).max()))


@pytest.mark.parametrize("case", "abc")
def test_cplx_linear_fp32_golden(golden, pkg, case):
    """cplx.linear forward + backward (fp32) vs the reference's outputs and autograd grads."""
    from gpu_util import T, N
    from cplxmodule_amd import cplx
    g = golden("linear")
    k = f"f32_{case}_"
    leaves = {n: T(g[k + n]).requires_grad_(True) for n in ("xr", "xi", "wr", "wi", "br", "bi")}
    y = cplx.linear(cplx.Cplx(leaves["xr"], leaves["xi"]), cplx.Cplx(leaves["wr"], leaves["wi"]),
                    cplx.Cplx(leaves["br"], leaves["bi"]))
    np.testing.assert_allclose(N(y.real), g[k + "y_naive_r"], **_scale_tol(g[k + "y_naive_r"], 1e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "y_naive_i"], **_scale_tol(g[k + "y_naive_i"], 1e-5))
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    for n, m in (("xr", "dxr"), ("xi", "dxi"), ("wr", "dwr"), ("wi", "dwi"), ("br", "dbr"), ("bi", "dbi")):
        np.testing.assert_allclose(N(leaves[n].grad), g[k + m], **_scale_tol(g[k + m], 1e-5))
    y = cplx.linear(cplx.Cplx(leaves["xr"], leaves["xi"]), cplx.Cplx(leaves["wr"], leaves["wi"]), None)
    np.testing.assert_allclose(N(y.real), g[k + "y_nobias_r"], **_scale_tol(g[k + "y_nobias_r"], 1e-5))
    # the reference's other two spellings (cplx.py:651-694), against THEIR OWN recorded outputs
    for algo in ("3m", "cat"):
        y = getattr(cplx, "linear_" + algo)(cplx.Cplx(leaves["xr"], leaves["xi"]), cplx.Cplx(leaves["wr"], leaves["wi"]),
                                            cplx.Cplx(leaves["br"], leaves["bi"]))
        np.testing.assert_allclose(N(y.real), g[k + f"y_{algo}_r"], **_scale_tol(g[k + f"y_{algo}_r"], 1e-5))
        np.testing.assert_allclose(N(y.imag), g[k + f"y_{algo}_i"], **_scale_tol(g[k + f"y_{algo}_i"], 1e-5))


def test_matmul_golden(golden, pkg):
    from gpu_util import T, N
    from cplxmodule_amd import cplx
    g = golden("linear")
    k = "f32_mm_"
    u = cplx.Cplx(T(g[k + "ur"]), T(g[k + "ui"]))
    v = cplx.Cplx(T(g[k + "vr"]), T(g[k + "vi"]))
    m = u @ v
    np.testing.assert_allclose(N(m.real), g[k + "mr"], **_scale_tol(g[k + "mr"], 1e-5))
    np.testing.assert_allclose(N(m.imag), g[k + "mi"], **_scale_tol(g[k + "mi"], 1e-5))
    m2 = u[0] @ v[0]
    np.testing.assert_allclose(N(m2.imag), g[k + "mi"][0], **_scale_tol(g[k + "mi"], 1e-5))


@pytest.mark.parametrize("batch,M,K,N_", [((3, 5), 17, 9, 6), ((70,), 130, 40, 129), ((1,), 1, 1, 1)])
def test_batched_matmul_and_grads(pkg, batch, M, K, N_):
    """Cplx.__matmul__ on [..., M, K] @ [..., K, N]: ONE batched launch; values and gradients against
    complex128 numpy."""
    from gpu_util import T, N
    from cplxmodule_amd import cplx
    rs = np.random.RandomState(M + K)
    u = rs.randn(*batch, M, K) + 1j * rs.randn(*batch, M, K)
    v = rs.randn(*batch, K, N_) + 1j * rs.randn(*batch, K, N_)
    g = rs.randn(*batch, M, N_) + 1j * rs.randn(*batch, M, N_)
    f = lambda a: T(np.ascontiguousarray(a).astype(np.float32)).requires_grad_(True)  # noqa: E731
    ur, ui, vr, vi = f(u.real), f(u.imag), f(v.real), f(v.imag)
    m = cplx.Cplx(ur, ui) @ cplx.Cplx(vr, vi)
    ref = u @ v
    assert m.shape == ref.shape
    np.testing.assert_allclose(N(m.real), ref.real, **_scale_tol(ref.real, 2e-5))
    np.testing.assert_allclose(N(m.imag), ref.imag, **_scale_tol(ref.imag, 2e-5))
    import torch
    torch.autograd.backward((m.real, m.imag), (T(g.real.astype(np.float32)), T(g.imag.astype(np.float32))))
    du = g @ np.conj(np.swapaxes(v, -1, -2))            # dU = G V^H, dV = U^H G (planar convention)
    dv = np.conj(np.swapaxes(u, -1, -2)) @ g
    for got, want in ((ur.grad, du.real), (ui.grad, du.imag), (vr.grad, dv.real), (vi.grad, dv.imag)):
        np.testing.assert_allclose(N(got), want, **_scale_tol(want, 3e-5))


@pytest.mark.parametrize("M,N_,K", [(128, 128, 32), (256, 384, 64), (200, 136, 96), (1, 5, 32),
                                    (130, 130, 160), (64, 64, 40), (33, 17, 7),
                                    (128, 128, 4096), (260, 132, 2048)])  # last two: split-K
@pytest.mark.parametrize("conj", (False, True))
def test_cgemm_bf16_vs_oracle(pkg, M, N_, K, conj):
    """bf16 MFMA path (and its generic fallback for K % 32 != 0): inputs rounded to bf16, the
    oracle multiplies the same rounded values in float64.  Asymmetric operands (G9)."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import ops
    rs = np.random.RandomState(M * 7 + N_ + K)
    ar, ai = bf16_round(rs.randn(M, K)), bf16_round(rs.randn(M, K) + 0.3)
    br, bi = bf16_round(rs.randn(N_, K) * 0.5 + 0.1), bf16_round(rs.randn(N_, K))
    bias = (rs.randn(N_).astype(np.float32), rs.randn(N_).astype(np.float32))
    a64 = ar.astype(np.float64) + 1j * ai
    b64 = br.astype(np.float64) + 1j * bi
    ref = a64 @ (b64.conj() if conj else b64).T + (bias[0] + 1j * bias[1])
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    cr, ci = ops.cgemm(q(ar), q(ai), (K, 1), q(br), q(bi), (K, 1), M, N_, K,
                       bias=(T(bias[0]), T(bias[1])), conj_b=conj, out_dtype=torch.float32)
    tol = 2e-6 * np.sqrt(K) * 4 + 1e-6
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(N(cr), ref.real, rtol=1e-5, atol=tol * scale)
    np.testing.assert_allclose(N(ci), ref.imag, rtol=1e-5, atol=tol * scale)
    cr16, ci16 = ops.cgemm(q(ar), q(ai), (K, 1), q(br), q(bi), (K, 1), M, N_, K, conj_b=conj,
                           out_dtype=torch.bfloat16)
    ref2 = ref - (bias[0] + 1j * bias[1])
    np.testing.assert_allclose(N(cr16), ref2.real, rtol=8e-3, atol=8e-3 * scale)
    np.testing.assert_allclose(N(ci16), ref2.imag, rtol=8e-3, atol=8e-3 * scale)


@pytest.mark.parametrize("ta,tb", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N_,K", [(256, 128, 64), (264, 136, 96), (512, 384, 160), (128, 128, 4096), (40, 24, 32)])
@pytest.mark.parametrize("conj", (False, True))
def test_cgemm_bf16_transposed_operands(pkg, M, N_, K, conj, ta, tb):
    """K-major ("T") operands through the ds_read_b64_tr_b16 path: A given as [K, M] and / or B
    as [K, N], addressed by strides; asymmetric data so a row / column swap cannot hide."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import ops
    rs = np.random.RandomState(M + 3 * N_ + K)
    ar, ai = bf16_round(rs.randn(M, K)), bf16_round(rs.randn(M, K) + 0.3)
    br, bi = bf16_round(rs.randn(N_, K) * 0.5 + 0.1), bf16_round(rs.randn(N_, K))
    ref = (ar.astype(np.float64) + 1j * ai) @ ((br.astype(np.float64) + 1j * bi).conj() if conj else
                                              (br.astype(np.float64) + 1j * bi)).T
    q = lambda a: T(np.ascontiguousarray(a), torch.bfloat16)  # noqa: E731
    A = (q(ar.T), q(ai.T), (1, M)) if ta else (q(ar), q(ai), (K, 1))
    Bm = (q(br.T), q(bi.T), (1, N_)) if tb else (q(br), q(bi), (K, 1))
    cr, ci = ops.cgemm(A[0], A[1], A[2], Bm[0], Bm[1], Bm[2], M, N_, K, conj_b=conj,
                       out_dtype=torch.float32)
    scale = float(np.abs(ref).max())
    tol = (2e-6 * np.sqrt(K) * 4 + 1e-6) * scale
    np.testing.assert_allclose(N(cr), ref.real, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(N(ci), ref.imag, rtol=1e-5, atol=tol)
    c = ops.rgemm(A[0], A[2], Bm[0], Bm[2], M, N_, K)
    rref = ar.astype(np.float64) @ br.astype(np.float64).T
    np.testing.assert_allclose(N(c), rref, rtol=1e-5, atol=1e-5 * np.abs(rref).max())


@pytest.mark.parametrize("lay", ["nn", "nt", "tt"])
@pytest.mark.parametrize("M,N_,K", [(256, 128, 64), (264, 136, 96), (40, 24, 32), (512, 384, 1024)])
@pytest.mark.parametrize("conj", (False, True))
def test_cgemm_gauss_3m(pkg, M, N_, K, conj, lay):
    """algo = 3M: t1 = Ar Br, t2 = Ai Bi', t3 = (Ar + Ai)(Br + Bi') with the operand sums rounded
    to bf16, re = t1 - t2, im = t3 - t1 - t2 (Bi' = -Bi for conj).  Checked against exactly that
    model in float64 (fp32-accumulation tolerance), and against the true product at the looser
    tolerance the extra bf16 rounding of the sums costs."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import ops
    rs = np.random.RandomState(M + 5 * N_ + K + conj)
    ar, ai = bf16_round(rs.randn(M, K)), bf16_round(rs.randn(M, K) + 0.3)
    br, bi = bf16_round(rs.randn(N_, K) * 0.5 + 0.1), bf16_round(rs.randn(N_, K))
    bias = (rs.randn(N_).astype(np.float32), rs.randn(N_).astype(np.float32))
    s = -1.0 if conj else 1.0
    f = lambda a: a.astype(np.float64)  # noqa: E731
    t1, t2 = f(ar) @ f(br).T, s * (f(ai) @ f(bi).T)
    t3 = f(bf16_round(ar + ai)) @ f(bf16_round(br + s * bi)).T
    mr, mi = t1 - t2 + bias[0], t3 - t1 - t2 + bias[1]
    q = lambda a: T(np.ascontiguousarray(a), torch.bfloat16)  # noqa: E731
    ta, tb = lay[0] == "t", lay[1] == "t"
    A = (q(ar.T), q(ai.T), (1, M)) if ta else (q(ar), q(ai), (K, 1))
    Bm = (q(br.T), q(bi.T), (1, N_)) if tb else (q(br), q(bi), (K, 1))
    cr, ci = ops.cgemm(A[0], A[1], A[2], Bm[0], Bm[1], Bm[2], M, N_, K, conj_b=conj,
                       bias=(T(bias[0]), T(bias[1])), out_dtype=torch.float32, algo=1)
    scale = float(max(np.abs(t1).max(), np.abs(t3).max()))
    tol = (2e-6 * np.sqrt(K) * 4 + 1e-6) * scale
    np.testing.assert_allclose(N(cr), mr, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(N(ci), mi, rtol=1e-5, atol=tol)
    true = (f(ar) + 1j * f(ai)) @ ((f(br) + 1j * s * f(bi))).T + (bias[0] + 1j * bias[1])
    err = np.abs((N(cr) + 1j * N(ci)) - true).max()
    assert err <= 2 ** -8 * np.sqrt(K) * 4 * 1.5 + 1e-3, err      # sums carry a 2^-9 relative rounding
    c16r, c16i = ops.cgemm(A[0], A[1], A[2], Bm[0], Bm[1], Bm[2], M, N_, K, conj_b=conj,
                           out_dtype=torch.bfloat16, algo=1)
    np.testing.assert_allclose(N(c16r), mr - bias[0], rtol=8e-3, atol=8e-3 * scale)
    np.testing.assert_allclose(N(c16i), mi - bias[1], rtol=8e-3, atol=8e-3 * scale)


def test_cgemm_gauss_rejects_what_it_cannot_do(pkg):
    """3M never degrades silently: float32 operands and strided (non-dense) operands are refused."""
    from gpu_util import T
    from cplxmodule_amd import ops
    from cplxmodule_amd._lib import CplxAmdError
    a = T(np.ones((64, 64), np.float32))
    with pytest.raises(CplxAmdError):
        ops.cgemm(a, a, (64, 1), a, a, (64, 1), 64, 64, 64, algo=1)
    b = a.bfloat16()
    with pytest.raises(CplxAmdError):
        ops.cgemm(b, b, (64, 1), b, b, (64, 1), 64, 64, 32, algo=1)     # lda != K


@pytest.mark.parametrize("B,I,O", [(96, 128, 96), (96, 128, 72), (50, 40, 24)])
def test_linear_3m_layer_fwd_bwd(pkg, B, I, O):
    """cplx.linear_3m (cplxmodule/cplx.py:669-694) against cplx.linear on bf16 activations:
    forward and all gradients agree to the bf16-sum rounding (shapes the 3M entry refuses run
    the 4M kernel at the layer level)."""
    from gpu_util import DEV
    from cplxmodule_amd import Cplx, cplx
    torch.manual_seed(1)
    w = Cplx(torch.randn(O, I, device=DEV, requires_grad=True), torch.randn(O, I, device=DEV, requires_grad=True))
    b = Cplx(torch.randn(O, device=DEV, requires_grad=True), torch.randn(O, device=DEV, requires_grad=True))
    x = Cplx(torch.randn(B, I, device=DEV).bfloat16().requires_grad_(True),
             torch.randn(B, I, device=DEV).bfloat16().requires_grad_(True))
    outs = []
    true_3m = lambda x_, w_, b_: cplx.linear_3m(x_, w_, b_, true_3m=True)  # noqa: E731
    # (the plain cplx.linear_3m call is routed to the 4M kernel for bf16: bit-identical to cplx.linear)
    y4, y3 = cplx.linear(x, w, b), cplx.linear_3m(x, w, b)
    assert torch.equal(y4.real, y3.real) and torch.equal(y4.imag, y3.imag)
    for fn in (cplx.linear, true_3m):
        for t in (w.real, w.imag, b.real, b.imag, x.real, x.imag):
            t.grad = None
        y = fn(x, w, b)
        torch.autograd.backward((y.real, y.imag), (torch.ones_like(y.real), 0.5 * torch.ones_like(y.imag)))
        outs.append([y.real.float(), y.imag.float()] + [t.grad.float() for t in (w.real, w.imag, b.real, b.imag, x.real, x.imag)])
    for a_, b_ in zip(*outs):
        scale = float(a_.abs().max())
        assert float((a_ - b_).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("M,N_,K", [(128, 128, 64), (70, 190, 96), (64, 64, 24), (128, 256, 4096)])
def test_rgemm_bf16_and_f32(pkg, M, N_, K):
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import ops
    rs = np.random.RandomState(5)
    a, b = bf16_round(rs.randn(M, K)), bf16_round(rs.randn(N_, K) + 0.2)
    em = rs.rand(M, N_).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    c = ops.rgemm(T(a, torch.bfloat16), (K, 1), T(b, torch.bfloat16), (K, 1), M, N_, K, emul=T(em))
    np.testing.assert_allclose(N(c), ref * em, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    c = ops.rgemm(T(a), (K, 1), T(b), (K, 1), M, N_, K)
    np.testing.assert_allclose(N(c), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    # transposed operands through strides (TN form)
    c = ops.rgemm(T(a.T.copy()), (1, M), T(b.T.copy()), (1, N_), M, N_, K)
    np.testing.assert_allclose(N(c), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


def test_lrt_cplx_linear_layer_golden(golden, pkg):
    """CplxLinearVD in training mode with the reference's recorded noise tape: forward, all
    gradients, eval mode, penalties, masks -- the cfg1 shapes (B=64, 128 -> 128, fp32)."""
    from gpu_util import T, N
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("lrt_linear")
    k = "f32_cplx_"
    layer = rel.CplxLinearVD(128, 128).to("cuda")
    sd = {"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
          "bias.imag": T(g[k + "bi"]), "log_sigma2": T(g[k + "ls2"])}
    layer.load_state_dict(sd)
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    tape = T(g[k + "tape"]) / np.float32(np.sqrt(2.0))
    layer.train()
    y = layer(cplx.Cplx(xr, xi), eps=cplx.Cplx(tape[0], tape[1]))
    tol = lambda r: dict(rtol=1e-5, atol=1e-5 * float(np.abs(r).max()))  # noqa: E731
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **tol(g[k + "yr"]))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **tol(g[k + "yi"]))
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad,
               dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), g[k + n], rtol=2e-5, atol=2e-5 * float(np.abs(g[k + n]).max()),
                                   err_msg=n)
    layer.eval()
    y = layer(cplx.Cplx(xr, xi))
    np.testing.assert_allclose(N(y.real), g[k + "yr_eval"], **tol(g[k + "yr_eval"]))


def test_lrt_clamp_boundary_layer(golden, pkg):
    from gpu_util import T, N
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("lrt_linear")
    k = "f32_cplx_"
    layer = rel.CplxLinearVD(128, 128).to("cuda")
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]),
                           "bias.real": T(g[k + "br"]), "bias.imag": T(g[k + "bi"]),
                           "log_sigma2": T(g[k + "clamp_ls2"])})
    xr, xi = T(g[k + "clamp_xr"]).requires_grad_(True), T(g[k + "clamp_xi"]).requires_grad_(True)
    tape = T(g[k + "clamp_tape"]) / np.float32(np.sqrt(2.0))
    y = layer(cplx.Cplx(xr, xi), eps=cplx.Cplx(tape[0], tape[1]))
    np.testing.assert_allclose(N(y.real), g[k + "clamp_yr"], rtol=1e-5, atol=1e-6)
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    ref = g[k + "clamp_dls2"]
    np.testing.assert_allclose(N(layer.log_sigma2.grad), ref, rtol=5e-5, atol=5e-5 * np.abs(ref).max())
    ref = g[k + "clamp_dxr"]
    np.testing.assert_allclose(N(xr.grad), ref, rtol=5e-5, atol=5e-5 * np.abs(ref).max())


def test_lrt_real_linear_layer_golden(golden, pkg):
    from gpu_util import T, N
    from cplxmodule_amd.nn import relevance as rel
    g = golden("lrt_linear")
    k = "f32_real_"
    layer = rel.LinearVD(128, 128).to("cuda")
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x = T(g[k + "x"]).requires_grad_(True)
    layer.train()
    y = layer(x, eps=T(g[k + "eps"]))
    np.testing.assert_allclose(N(y), g[k + "y"], rtol=1e-5, atol=1e-5 * np.abs(g[k + "y"]).max())
    (y * T(g[k + "g"])).sum().backward()
    for n, t in dict(dx=x.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad).items():
        np.testing.assert_allclose(N(t), g[k + n], rtol=2e-5, atol=2e-5 * np.abs(g[k + n]).max(), err_msg=n)


@pytest.mark.parametrize("kind,cls", [("cplx_vd", "CplxLinearVD"), ("cplx_ard", "CplxLinearARD"),
                                      ("real_vd", "LinearVD"), ("real_ard", "LinearARD")])
def test_layer_penalty_api(golden, pkg, kind, cls):
    """.penalty / penalties(sum|mean) / relevance / compute_ard_masks / sparsity through the
    module API, values and gradients vs the golden vectors."""
    from gpu_util import T, N
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.utils import sparsity
    g = golden("penalty")
    O, I = g["f32_ls2"].shape
    layer = getattr(rel, cls)(I, O, bias=False).to("cuda")
    if kind.startswith("cplx"):
        layer.load_state_dict({"weight.real": T(g["f32_wr"]), "weight.imag": T(g["f32_wi"]),
                               "log_sigma2": T(g["f32_ls2"])})
    else:
        layer.load_state_dict({"weight": T(g["f32_wr"]), "log_sigma2": T(g["f32_ls2"])})
    ref = g[f"f32_{kind}_penalty"]
    fin = np.isfinite(ref)
    np.testing.assert_allclose(N(layer.penalty)[fin], ref[fin], rtol=1e-5, atol=2e-6)
    tot = sum(rel.penalties(layer, reduction="sum"))
    np.testing.assert_allclose(float(tot), float(g[f"f32_{kind}_sum"]), rtol=1e-5)
    mean = sum(rel.penalties(layer, reduction="mean"))
    np.testing.assert_allclose(float(mean), float(g[f"f32_{kind}_mean"]), rtol=1e-5)
    (0.5 * tot).backward()
    ref = 0.5 * g[f"f32_{kind}_sum_dls2"]
    np.testing.assert_allclose(N(layer.log_sigma2.grad), ref, rtol=2e-5, atol=1e-6)
    masks = rel.compute_ard_masks(layer, threshold=1.0)
    assert list(masks) == ["mask"]
    np.testing.assert_array_equal(N(masks["mask"]), g[f"f32_{kind}_mask_1.0"])
    sp = sparsity(layer, threshold=1.0)
    assert 0.0 < sp < 1.0
    with pytest.raises(ValueError):
        list(rel.named_penalties(layer, reduction="max"))


def test_bf16_layer_vs_oracle(pkg):
    """CplxLinearVD with bf16 activations (the cfg2 numerics at a small size): the oracle runs
    on the bf16-rounded operands in float64; tolerance = bf16 output rounding."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    rs = np.random.RandomState(21)
    B, I, O = 256, 128, 192
    layer = rel.CplxLinearVD(I, O).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-8, -2)
    wr, wi = bf16_round(N(layer.weight.real)), bf16_round(N(layer.weight.imag))
    xr, xi = bf16_round(rs.randn(B, I) * 0.7), bf16_round(rs.randn(B, I) * 0.7)
    er, ei = bf16_round(rs.randn(B, O) * 0.7), bf16_round(rs.randn(B, O) * 0.7)
    gr, gi = bf16_round(rs.randn(B, O)), bf16_round(rs.randn(B, O))
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    txr, txi = q(xr).requires_grad_(True), q(xi).requires_grad_(True)
    y = layer(cplx.Cplx(txr, txi), eps=cplx.Cplx(q(er), q(ei)))
    assert y.real.dtype == torch.bfloat16
    ls2 = N(layer.log_sigma2).astype(np.float64)
    S16 = bf16_round(np.exp(N(layer.log_sigma2)))
    a16 = bf16_round(xr * xr + xi * xi)
    f = np.float64
    br, bi = N(layer.bias.real).astype(f), N(layer.bias.imag).astype(f)
    mur, mui = orc.cplx_linear(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br, bi)
    s2 = a16.astype(f) @ S16.astype(f).T
    sd = np.sqrt(np.maximum(s2, 1e-8))
    ref_r, ref_i = mur + er * sd, mui + ei * sd
    np.testing.assert_allclose(N(y.real), ref_r, rtol=1e-2, atol=1e-2 * np.abs(ref_r).max())
    np.testing.assert_allclose(N(y.imag), ref_i, rtol=1e-2, atol=1e-2 * np.abs(ref_i).max())
    ((y.real * q(gr)).sum() + (y.imag * q(gi)).sum()).backward()
    bw = orc.lrt_cplx_linear_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f),
                                 wi.astype(f), ls2, er.astype(f), ei.astype(f))
    for n, t in dict(dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad, dxr=txr.grad,
                     dbr=layer.bias.real.grad, dlog_sigma2=layer.log_sigma2.grad).items():
        ref = bw[n]
        np.testing.assert_allclose(N(t), ref, rtol=3e-2, atol=3e-2 * np.abs(ref).max(), err_msg=n)


def test_philox_layer_statistics(pkg):
    """Default (in-kernel Philox) noise: y - mu has the right variance and differs per call;
    backward regenerates the same noise (gradient wrt log_sigma2 matches a finite difference
    of the realised noise)."""
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(64, 96).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.fill_(-2.0)
    x = cplx.Cplx(torch.randn(4096, 64, device="cuda"), torch.randn(4096, 64, device="cuda"))
    layer.eval()
    mu = layer(x)
    layer.train()
    y1, y2 = layer(x), layer(x)
    d1 = torch.stack([y1.real - mu.real, y1.imag - mu.imag])
    s2 = ((x.real ** 2 + x.imag ** 2) @ torch.exp(layer.log_sigma2).T)
    ratio = (d1[0] ** 2 + d1[1] ** 2).mean() / s2.mean()
    assert abs(float(ratio) - 1.0) < 0.02          # E|eps|^2 = 1
    assert float((y1.real - y2.real).abs().mean()) > 1e-3
    assert abs(float(d1[0].mean())) < 5e-3 * float(s2.mean().sqrt())


@pytest.mark.parametrize("M,N_,K", [(256, 10, 1568), (10, 1568, 256), (64, 64, 1000), (70, 130, 4100), (3, 5, 777)])
@pytest.mark.parametrize("conj", (False, True))
def test_cgemm_f32_generic_split_k(pkg, M, N_, K, conj):
    """Few output tiles + long K: the exact-f32 kernel splits K into slabs (incl. a K that is not a
    multiple of the 16-wide K tile, and splits whose range is empty) -- against float64."""
    from gpu_util import T, N
    from cplxmodule_amd import ops
    rs = np.random.RandomState(M + N_ + K)
    ar, ai = rs.randn(M, K).astype(np.float32), rs.randn(M, K).astype(np.float32)
    br, bi = rs.randn(N_, K).astype(np.float32), rs.randn(N_, K).astype(np.float32)
    bias = (rs.randn(N_).astype(np.float32), rs.randn(N_).astype(np.float32))
    f = np.float64
    b64 = br.astype(f) + 1j * bi
    ref = (ar.astype(f) + 1j * ai) @ (b64.conj() if conj else b64).T + (bias[0] + 1j * bias[1])
    cr, ci = ops.cgemm(T(ar), T(ai), (K, 1), T(br), T(bi), (K, 1), M, N_, K, bias=(T(bias[0]), T(bias[1])),
                       conj_b=conj)
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(N(cr), ref.real, rtol=1e-5, atol=2e-6 * np.sqrt(K) * scale / 10)
    np.testing.assert_allclose(N(ci), ref.imag, rtol=1e-5, atol=2e-6 * np.sqrt(K) * scale / 10)
    # K-major B operand (the dgrad layout) and a real GEMM with emul through the same path
    cr2, ci2 = ops.cgemm(T(ar), T(ai), (K, 1), T(np.ascontiguousarray(br.T)), T(np.ascontiguousarray(bi.T)),
                         (1, N_), M, N_, K, conj_b=conj)
    ref2 = ref - (bias[0] + 1j * bias[1])
    np.testing.assert_allclose(N(cr2), ref2.real, rtol=1e-5, atol=2e-6 * np.sqrt(K) * scale / 10)
    em = rs.rand(M, N_).astype(np.float32)
    c = ops.rgemm(T(ar), (K, 1), T(br), (K, 1), M, N_, K, emul=T(em))
    rref = (ar.astype(f) @ br.astype(f).T) * em
    np.testing.assert_allclose(N(c), rref, rtol=1e-5, atol=2e-6 * np.sqrt(K) * np.abs(rref).max() / 10)
