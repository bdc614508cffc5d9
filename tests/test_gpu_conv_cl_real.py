"""Real-valued channels-last conv kernels (csrc/conv_cl_real.hip, conv_cl_wgrad_real.hip) and the local-
reparameterization conv layers that run on them + the complex ones end to end (mean conv, variance conv, noise
injection, all gradients) against the float64 numpy oracle on the bf16-rounded operands, with a supplied noise tape."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_cl():
    from cplxmodule_amd import conv
    old = conv._CL_FORCE
    conv._CL_FORCE = True
    yield
    conv._CL_FORCE = old


CASES = {
    "same": dict(B=2, Ci=64, Co=64, H=8, W=32, padding=1, dilation=1),
    "valid": dict(B=3, Ci=64, Co=128, H=9, W=64, padding=0, dilation=1),
    "dil2_half_pad": dict(B=2, Ci=128, Co=64, H=11, W=32, padding=(1, 2), dilation=2),
    "many_tiles": dict(B=2, Ci=64, Co=64, H=40, W=96, padding=1, dilation=1),
    "width_28": dict(B=3, Ci=64, Co=64, H=28, W=28, padding=1, dilation=1),
    "width_50_valid_dil2": dict(B=2, Ci=64, Co=64, H=12, W=50, padding=(0, 1), dilation=(1, 2)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_real_conv_cl_vs_oracle(case):
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import conv
    cfg = CASES[case]
    rs = np.random.RandomState(len(case) + 1)
    B, Ci, Co, H, W = cfg["B"], cfg["Ci"], cfg["Co"], cfg["H"], cfg["W"]
    x, w = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(Co, Ci, 3, 3) * 0.1)
    b = rs.randn(Co).astype(np.float32)
    tx = T(x, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tw, tb = T(w).requires_grad_(True), T(b).requires_grad_(True)
    y = conv.RealConv2dFn.apply(tx, tw, tb, 1, cfg["padding"], cfg["dilation"], 1)
    assert y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()
    f = np.float64
    kw = dict(padding=cfg["padding"], dilation=cfg["dilation"])
    ref = orc.real_conv2d(x.astype(f), w.astype(f), **kw) + b.astype(f)[None, :, None, None]
    np.testing.assert_allclose(N(y), ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max())
    g = bf16_round(rs.randn(*ref.shape))
    (y * T(g, torch.bfloat16)).sum().backward()
    dx, dw = orc.real_conv2d_bwd(g.astype(f), x.astype(f), w.astype(f), **kw)
    for n, t, r in (("dx", tx.grad, dx), ("dw", tw.grad, dw), ("db", tb.grad, g.astype(f).sum((0, 2, 3)))):
        np.testing.assert_allclose(N(t), r, rtol=2e-2, atol=2e-2 * np.abs(r).max(), err_msg=n)


@pytest.mark.parametrize("padding", [1, 0])
def test_lrt_cplx_conv_layer_channels_last(padding):
    """CplxConv2dVD in training mode on channels-last images: forward with the supplied noise tape and every gradient
    (input, weight, bias, log_sigma2) against the oracle."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import Cplx, conv
    from cplxmodule_amd.nn import relevance as rel
    rs = np.random.RandomState(7 + padding)
    B, C, H, W = 2, 64, 10, 32
    layer = rel.CplxConv2dVD(C, C, 3, padding=padding).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-6, -1)
    xr, xi = bf16_round(rs.randn(B, C, H, W)), bf16_round(rs.randn(B, C, H, W))
    cl = lambda a: T(a, torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    txr, txi = cl(xr).requires_grad_(True), cl(xi).requires_grad_(True)
    Ho, Wo = H + 2 * padding - 2, W + 2 * padding - 2
    er, ei = bf16_round(rs.randn(B, C, Ho, Wo) / np.sqrt(2)), bf16_round(rs.randn(B, C, Ho, Wo) / np.sqrt(2))
    layer.train()
    geom, _ = conv._geom(txr.shape, layer.weight.real.shape, 1, padding, 1, 1)
    assert conv._cl_layer_ok(geom, txr, txi)
    y = layer(Cplx(txr, txi), eps=Cplx(T(er, torch.bfloat16), T(ei, torch.bfloat16)))
    assert y.real.is_contiguous(memory_format=torch.channels_last) and not y.real.is_contiguous()
    n = lambda v: v.detach().float().cpu().numpy().astype(np.float64)  # noqa: E731
    f = np.float64
    wr, wi = bf16_round(n(layer.weight.real)), bf16_round(n(layer.weight.imag))
    br, bi, ls2 = n(layer.bias.real), n(layer.bias.imag), n(layer.log_sigma2)
    yr, yi, _ = orc.lrt_cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br, bi, ls2, er.astype(f), ei.astype(f),
                                    padding=padding)
    np.testing.assert_allclose(N(y.real), yr, rtol=2e-2, atol=2e-2 * np.abs(yr).max())
    np.testing.assert_allclose(N(y.imag), yi, rtol=2e-2, atol=2e-2 * np.abs(yi).max())
    gr, gi = bf16_round(rs.randn(*yr.shape)), bf16_round(rs.randn(*yr.shape))
    torch.autograd.backward((y.real, y.imag), (cl(gr), cl(gi)))
    bw = orc.lrt_cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), ls2,
                                 er.astype(f), ei.astype(f), padding=padding)
    got = dict(dxr=txr.grad, dxi=txi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad, dbr=layer.bias.real.grad,
               dbi=layer.bias.imag.grad, dlog_sigma2=layer.log_sigma2.grad)
    for k, t in got.items():
        np.testing.assert_allclose(N(t), bw[k], rtol=3e-2, atol=3e-2 * np.abs(bw[k]).max(), err_msg=k)


def test_lrt_real_conv_layer_channels_last():
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd.nn import relevance as rel
    rs = np.random.RandomState(17)
    B, C, H, W = 2, 64, 9, 64
    layer = rel.Conv2dVD(C, C, 3, padding=1).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-6, -1)
    x = bf16_round(rs.randn(B, C, H, W))
    tx = T(x, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    eps = bf16_round(rs.randn(B, C, H, W))
    layer.train()
    y = layer(tx, eps=T(eps, torch.bfloat16))
    assert y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()
    n = lambda v: v.detach().float().cpu().numpy().astype(np.float64)  # noqa: E731
    f = np.float64
    w, b, ls2 = bf16_round(n(layer.weight)).astype(f), n(layer.bias), n(layer.log_sigma2)
    ref, _ = orc.lrt_real_conv2d(x.astype(f), w, b, ls2, eps.astype(f), padding=1)
    np.testing.assert_allclose(N(y), ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())
    g = bf16_round(rs.randn(*ref.shape))
    (y * T(g, torch.bfloat16).contiguous(memory_format=torch.channels_last)).sum().backward()
    # gradients through the oracle's pieces: mean conv + variance conv (real/base.py:116-163)
    S = np.exp(ls2)
    s2 = orc.real_conv2d(x.astype(f) ** 2, S, padding=1)
    sd = np.sqrt(np.maximum(s2, 1e-8))
    gs2 = np.where(s2 >= 1e-8, g.astype(f) * eps.astype(f) * 0.5 / sd, 0.0)
    dx_mu, dw = orc.real_conv2d_bwd(g.astype(f), x.astype(f), w, padding=1)
    ga, dS = orc.real_conv2d_bwd(gs2, x.astype(f) ** 2, S, padding=1)
    want = dict(dx=dx_mu + 2 * x.astype(f) * ga, dw=dw, db=g.astype(f).sum((0, 2, 3)), dls2=dS * S)
    got = dict(dx=tx.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad)
    for k, t in got.items():
        np.testing.assert_allclose(N(t), want[k], rtol=3e-2, atol=3e-2 * np.abs(want[k]).max(), err_msg=k)
