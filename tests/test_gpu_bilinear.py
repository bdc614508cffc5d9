"""SURVEY 8(f) row 4 on the GPU: cplx.bilinear / CplxBilinear, the bilinear VD / ARD layers (complex
and real) -- against the reference's golden vectors and the numpy oracle."""
import numpy as np
import pytest
import torch

import oracle.cplx_oracle as orc
from gpu_util import DEV, T, N

pytestmark = pytest.mark.gpu


def _tol(ref, r=2e-5):
    return dict(rtol=r, atol=r * float(np.abs(ref).max()))


@pytest.mark.parametrize("mode", ["conj", "plain"])
def test_bilinear_golden(golden, mode):
    from cplxmodule_amd import Cplx, cplx
    g = golden("bilinear")
    k = f"f32_fn_{mode}_"
    names = ("x1r", "x1i", "x2r", "x2i", "wr", "wi", "br", "bi")
    t = {n: T(g[k + n]).requires_grad_(True) for n in names}
    y = cplx.bilinear(Cplx(t["x1r"], t["x1i"]), Cplx(t["x2r"], t["x2i"]), Cplx(t["wr"], t["wi"]),
                      Cplx(t["br"], t["bi"]), conjugate=mode == "conj")
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"], 1e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"], 1e-5))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    for n in names:
        np.testing.assert_allclose(N(t[n].grad), g[k + "d" + n], **_tol(g[k + "d" + n]), err_msg=n)


def test_bilinear_layer_leading_dims(golden):
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import CplxBilinear
    g = golden("bilinear")
    k = "f32_layer_"
    O, I1, I2 = g[k + "wr"].shape
    layer = CplxBilinear(I1, I2, O, bias=False).to(DEV)
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"])})
    y = layer(Cplx(T(g[k + "x1r"]), T(g[k + "x1i"])), Cplx(T(g[k + "x2r"]), T(g[k + "x2i"])))
    assert y.shape == g[k + "yr"].shape
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"], 1e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"], 1e-5))
    assert "conjugate=True" in repr(layer)


def test_cplx_bilinear_vd_golden(golden):
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("bilinear")
    k = "f32_vd_"
    O, I1, I2 = g[k + "wr"].shape
    layer = rel.CplxBilinearVD(I1, I2, O).to(DEV)
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]),
                           "bias.real": T(g[k + "br"]), "bias.imag": T(g[k + "bi"]),
                           "log_sigma2": T(g[k + "ls2"])})
    x = {n: T(g[k + n]).requires_grad_(True) for n in ("x1r", "x1i", "x2r", "x2i")}
    tape = T(g[k + "tape"]) / np.float32(np.sqrt(2.0))
    layer.train()
    y = layer(Cplx(x["x1r"], x["x1i"]), Cplx(x["x2r"], x["x2i"]), eps=Cplx(tape[0], tape[1]))
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"], 1e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"], 1e-5))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    got = dict(dx1r=x["x1r"].grad, dx1i=x["x1i"].grad, dx2r=x["x2r"].grad, dx2i=x["x2i"].grad,
               dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad, dbr=layer.bias.real.grad,
               dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), g[k + n], **_tol(g[k + n], 3e-5), err_msg=n)
    layer.eval()
    y = layer(Cplx(x["x1r"], x["x1i"]), Cplx(x["x2r"], x["x2i"]))
    np.testing.assert_allclose(N(y.real), g[k + "yr_eval"], **_tol(g[k + "yr_eval"], 1e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "yi_eval"], **_tol(g[k + "yi_eval"], 1e-5))
    np.testing.assert_allclose(N(layer.penalty), g[k + "pen"], rtol=1e-5, atol=2e-6)
    assert layer.penalty.shape == (O, I1, I2)
    mask = layer.relevance(threshold=-2.0)
    assert np.array_equal(N(mask), orc.relevance_mask(np.float32(-2.0), g[k + "ls2"], g[k + "wr"], g[k + "wi"]))


def test_real_bilinear_vd_golden(golden):
    from cplxmodule_amd.nn import relevance as rel
    g = golden("bilinear")
    k = "f32_real_"
    O, I1, I2 = g[k + "w"].shape
    layer = rel.BilinearVD(I1, I2, O).to(DEV)
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x1, x2 = T(g[k + "x1"]).requires_grad_(True), T(g[k + "x2"]).requires_grad_(True)
    layer.train()
    y = layer(x1, x2, eps=T(g[k + "eps"]))
    np.testing.assert_allclose(N(y), g[k + "y"], **_tol(g[k + "y"], 1e-5))
    (y * T(g[k + "g"])).sum().backward()
    got = dict(dx1=x1.grad, dx2=x2.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), g[k + n], **_tol(g[k + n], 3e-5), err_msg=n)
    layer.eval()
    np.testing.assert_allclose(N(layer(x1, x2)), g[k + "y_eval"], **_tol(g[k + "y_eval"], 1e-5))
    assert isinstance(rel.BilinearARD(3, 4, 5), rel.BilinearVD)


@pytest.mark.parametrize("B,I1,I2,O", [(1, 1, 1, 1), (5, 3, 300, 2), (37, 300, 5, 3), (64, 17, 33, 65),
                                        (256, 64, 64, 32)])
@pytest.mark.parametrize("conj", [True, False])
def test_bilinear_shapes_vs_oracle(B, I1, I2, O, conj):
    """every group size of the reduction kernel, ragged sizes, the split between GEMM and reduction"""
    from cplxmodule_amd import Cplx, cplx
    rng = np.random.default_rng(B + I1)
    a = [rng.standard_normal(s).astype(np.float32) for s in ((B, I1), (B, I1), (B, I2), (B, I2))]
    w = [(0.3 * rng.standard_normal((O, I1, I2))).astype(np.float32) for _ in range(2)]
    b = [rng.standard_normal(O).astype(np.float32) for _ in range(2)]
    gr, gi = rng.standard_normal((B, O)).astype(np.float32), rng.standard_normal((B, O)).astype(np.float32)
    t = [T(v).requires_grad_(True) for v in a + w + b]
    y = cplx.bilinear(Cplx(t[0], t[1]), Cplx(t[2], t[3]), Cplx(t[4], t[5]), Cplx(t[6], t[7]), conjugate=conj)
    a64 = [v.astype(np.float64) for v in a + w + b]
    yr, yi = orc.cplx_bilinear(*a64, conjugate=conj)
    np.testing.assert_allclose(N(y.real), yr, **_tol(yr))
    np.testing.assert_allclose(N(y.imag), yi, **_tol(yi))
    torch.autograd.backward((y.real, y.imag), (T(gr), T(gi)))
    bw = orc.cplx_bilinear_bwd(gr.astype(np.float64), gi.astype(np.float64), *a64[:6], conjugate=conj)
    for n, v in zip(("dx1r", "dx1i", "dx2r", "dx2i", "dwr", "dwi", "dbr", "dbi"), t):
        np.testing.assert_allclose(N(v.grad), bw[n], **_tol(bw[n], 5e-5), err_msg=n)


def test_bilinear_vd_bf16_and_philox():
    """bf16 activations through the MFMA GEMMs; in-kernel noise: eval is noise-free, training noise
    has the variance the layer predicts, the same Philox offset replays the same draw."""
    from cplxmodule_amd import Cplx, cplx
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd.nn.relevance import extensions as ext
    torch.manual_seed(3)
    B, I1, I2, O = 512, 32, 64, 16
    layer = rel.CplxBilinearARD(I1, I2, O).to(DEV)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-6, -3)
    x1, x2 = cplx.randn(B, I1, device=DEV), cplx.randn(B, I2, device=DEV)
    a = [N(p).astype(np.float64) for p in (x1.real, x1.imag, x2.real, x2.imag, layer.weight.real,
                                            layer.weight.imag, layer.bias.real, layer.bias.imag)]
    mur, mui = orc.cplx_bilinear(*a)
    layer.eval()
    y = layer(x1, x2)
    np.testing.assert_allclose(N(y.real), mur, **_tol(mur))
    # bf16 activations
    h1, h2 = Cplx(x1.real.bfloat16(), x1.imag.bfloat16()), Cplx(x2.real.bfloat16(), x2.imag.bfloat16())
    yb = layer(h1, h2)
    assert yb.real.dtype == torch.bfloat16
    np.testing.assert_allclose(N(yb.real), mur, rtol=0.05, atol=0.05 * np.abs(mur).max())
    # training noise statistics (float32)
    layer.train()
    s2 = orc.real_bilinear(a[0] ** 2 + a[1] ** 2, a[2] ** 2 + a[3] ** 2, np.exp(N(layer.log_sigma2).astype(np.float64)))
    y1 = layer(x1, x2)
    zr, zi = (N(y1.real) - mur) / np.sqrt(s2), (N(y1.imag) - mui) / np.sqrt(s2)
    n = zr.size
    assert abs(zr.mean()) < 5 / np.sqrt(n) and abs(zi.mean()) < 5 / np.sqrt(n)
    assert abs(zr.var() - 0.5) < 0.05 and abs(zi.var() - 0.5) < 0.05
    # bf16 training step runs end to end, gradients finite
    h1 = Cplx(h1.real.requires_grad_(True), h1.imag.requires_grad_(True))
    yb = layer(h1, h2)
    (yb.real.float().square().sum() + yb.imag.float().square().sum()).backward()
    for p in (h1.real.grad, layer.weight.real.grad, layer.log_sigma2.grad, layer.bias.imag.grad):
        assert p is not None and bool(torch.isfinite(p.float()).all())
    # extension penalties on the bilinear layer
    for cls, kind in ((ext.CplxBilinearVDApprox, "cplx_vd_approx"), (ext.CplxBilinearVDScaleFree, "cplx_vd_scalefree")):
        m = cls(I1, I2, O).to(DEV)
        ref = orc.penalty(kind, N(m.log_sigma2), N(m.weight.real), N(m.weight.imag))
        np.testing.assert_allclose(N(m.penalty), ref, rtol=1e-4, atol=1e-5)


def test_bilinear_empty_batch():
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    layer = rel.CplxBilinearVD(4, 6, 3).to(DEV)
    x1, x2 = cplx.randn(0, 4, device=DEV), cplx.randn(0, 6, device=DEV)
    y = layer(x1, x2)
    assert y.shape == (0, 3)


def test_masked_bilinear_and_conv1d_layers():
    """the masked counterparts added with the bilinear family: weight * mask feeds the same kernels"""
    from cplxmodule_amd import Cplx, cplx
    from cplxmodule_amd.nn import masked
    torch.manual_seed(5)
    m = masked.CplxBilinearMasked(6, 7, 4).to(DEV)
    x1, x2 = cplx.randn(9, 6, device=DEV), cplx.randn(9, 7, device=DEV)
    with pytest.raises(RuntimeError):
        m(x1, x2)
    mask = (torch.rand(4, 6, 7, device=DEV) > 0.5).float()
    m.mask = mask
    y = m(x1, x2)
    w = m.weight
    ref = cplx.bilinear(x1, x2, Cplx(w.real * mask, w.imag * mask), m.bias, True)
    assert torch.equal(y.real, ref.real) and torch.equal(y.imag, ref.imag)
    a = [N(p).astype(np.float64) for p in (x1.real, x1.imag, x2.real, x2.imag, w.real * mask, w.imag * mask,
                                            m.bias.real, m.bias.imag)]
    yr, _ = orc.cplx_bilinear(*a)
    np.testing.assert_allclose(N(y.real), yr, **_tol(yr))
    (n_re, dropped), _ = m.sparsity()
    assert dropped == float(mask.numel() - mask.sum().item())

    r = masked.BilinearMasked(6, 7, 4).to(DEV)
    r.mask = mask
    u, v = torch.randn(9, 6, device=DEV), torch.randn(9, 7, device=DEV)
    ref = orc.real_bilinear(N(u).astype(np.float64), N(v).astype(np.float64),
                            N(r.weight * mask).astype(np.float64), N(r.bias).astype(np.float64))
    np.testing.assert_allclose(N(r(u, v)), ref, **_tol(ref))

    c = masked.CplxConv1dMasked(3, 5, 3, padding=1).to(DEV)
    cm = (torch.rand(5, 3, 3, device=DEV) > 0.4).float()
    c.mask = cm
    z = cplx.randn(2, 3, 11, device=DEV)
    ref = cplx.conv1d(z, Cplx(c.weight.real * cm, c.weight.imag * cm), c.bias, 1, 1)
    out = c(z)
    assert torch.equal(out.real, ref.real) and out.shape == (2, 5, 11)

    rc = masked.Conv1dMasked(3, 5, 3, stride=2).to(DEV)
    rc.mask = cm
    xin = torch.randn(2, 3, 11, device=DEV)
    ref = torch.nn.functional.conv1d(xin.cpu().double(), (rc.weight * cm).detach().cpu().double(),
                                     rc.bias.detach().cpu().double(), stride=2).numpy()
    np.testing.assert_allclose(N(rc(xin)), ref, **_tol(ref))
