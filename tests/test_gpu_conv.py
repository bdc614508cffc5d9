"""Implicit-GEMM complex / real conv kernels vs the reference's outputs and autograd grads."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc
from oracle.gen_golden_cases import CONV_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rows-forced", "size-gated"])
def conv_kernel_choice(request):
    """Every test runs twice: with the channels-last ("rows") kernels forced wherever the geometry
    allows (they are normally reserved for layers with enough work), and with the default choice,
    under which these small shapes take the gather kernels."""
    from cplxmodule_amd import conv
    old = conv._ROWS_FORCE
    conv._ROWS_FORCE = request.param == "rows-forced"
    yield request.param
    conv._ROWS_FORCE = old


def _tol(ref, r=2e-5):
    return dict(rtol=r, atol=r * float(np.abs(ref).max()))


@pytest.mark.parametrize("case", list(CONV_CASES))
def test_cplx_conv2d_layer_golden(golden, case):
    from gpu_util import T, N
    from cplxmodule_amd import Cplx, nn
    g = golden("conv")
    B, Ci, Co, H, W, ks, st, pd, dl, gp, mode = CONV_CASES[case]
    k = f"f32_{case}_"
    layer = nn.CplxConv2d(Ci, Co, ks, stride=st, padding=pd, dilation=dl, groups=gp,
                          padding_mode=mode).to("cuda")
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]),
                           "bias.real": T(g[k + "br"]), "bias.imag": T(g[k + "bi"])})
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    y = layer(Cplx(xr, xi))
    assert tuple(y.shape) == g[k + "yr"].shape
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"]))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"]))
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad,
               dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), g[k + n], **_tol(g[k + n], 5e-5), err_msg=n)


def test_conv_errors():
    from cplxmodule_amd import Cplx, nn, cplx
    with pytest.raises(ValueError):
        nn.CplxConv2d(3, 4, 3, groups=2)
    with pytest.raises(ValueError):
        cplx.conv2d(Cplx(torch.zeros(1, 1, 4, 4, device="cuda")), Cplx(torch.zeros(1, 1, 3, 3, device="cuda")),
                    padding_mode="reflect")
    from cplxmodule_amd.nn import relevance as rel
    with pytest.raises(ValueError):
        rel.CplxConv2dVD(2, 2, 3, padding_mode="circular")


def test_lrt_cplx_conv_golden(golden):
    from gpu_util import T, N
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("conv")
    k = "f32_lrtc_"
    layer = rel.CplxConv2dVD(3, 4, 3, stride=1, padding=1).to("cuda")
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]),
                           "bias.real": T(g[k + "br"]), "bias.imag": T(g[k + "bi"]),
                           "log_sigma2": T(g[k + "ls2"])})
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    tape = T(g[k + "tape"]) / np.float32(np.sqrt(2.0))
    layer.train()
    y = layer(Cplx(xr, xi), eps=Cplx(tape[0], tape[1]))
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"]))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"]))
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad,
               dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), g[k + n], **_tol(g[k + n], 5e-5), err_msg=n)
    tot = sum(rel.penalties(layer))
    np.testing.assert_allclose(float(tot), float(g[k + "penalty_sum"]), rtol=1e-5)
    layer.eval()
    y0 = layer(Cplx(xr, xi))
    mur, mui = orc.cplx_conv2d(g[k + "xr"], g[k + "xi"], g[k + "wr"], g[k + "wi"], g[k + "br"], g[k + "bi"],
                               stride=1, padding=1)
    np.testing.assert_allclose(N(y0.real), mur, **_tol(mur))


def test_lrt_real_conv_golden(golden):
    from gpu_util import T, N
    from cplxmodule_amd.nn import relevance as rel
    g = golden("conv")
    k = "f32_lrtr_"
    layer = rel.Conv2dVD(3, 4, 3, stride=2, padding=1).to("cuda")
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x = T(g[k + "x"]).requires_grad_(True)
    layer.train()
    y = layer(x, eps=T(g[k + "eps"]))
    np.testing.assert_allclose(N(y), g[k + "y"], **_tol(g[k + "y"]))
    (y * T(g[k + "g"])).sum().backward()
    for n, t in dict(dx=x.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad).items():
        np.testing.assert_allclose(N(t), g[k + n], **_tol(g[k + n], 5e-5), err_msg=n)


def test_conv_bf16_and_larger_vs_oracle():
    """A larger case (several tiles along every GEMM dim, split-K wgrad) in bf16: the oracle gets
    the bf16-rounded operands; linearity in x checks the full-size property cheaply."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import Cplx, cplx
    rs = np.random.RandomState(3)
    B, Ci, Co, H, W = 3, 16, 72, 20, 19
    xr, xi = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(B, Ci, H, W))
    wr, wi = bf16_round(rs.randn(Co, Ci, 3, 3) * 0.1), bf16_round(rs.randn(Co, Ci, 3, 3) * 0.1)
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    txr, txi = q(xr).requires_grad_(True), q(xi).requires_grad_(True)
    twr, twi = T(wr).requires_grad_(True), T(wi).requires_grad_(True)
    y = cplx.conv2d(Cplx(txr, txi), Cplx(twr, twi), None, stride=1, padding=1)
    f = np.float64
    yr, yi = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), padding=1)
    np.testing.assert_allclose(N(y.real), yr, rtol=1e-2, atol=1e-2 * np.abs(yr).max())
    np.testing.assert_allclose(N(y.imag), yi, rtol=1e-2, atol=1e-2 * np.abs(yi).max())
    gr, gi = bf16_round(rs.randn(*yr.shape)), bf16_round(rs.randn(*yr.shape))
    ((y.real * q(gr)).sum() + (y.imag * q(gi)).sum()).backward()
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f),
                             wi.astype(f), padding=1, has_bias=False)
    for n, t in dict(dxr=txr.grad, dxi=txi.grad, dwr=twr.grad, dwi=twi.grad).items():
        np.testing.assert_allclose(N(t), bw[n], rtol=2e-2, atol=2e-2 * np.abs(bw[n]).max(), err_msg=n)


@pytest.mark.parametrize("cfg", [
    dict(B=2, Ci=32, Co=64, H=18, W=21, k=3, stride=1, padding=1, dilation=1, groups=1),
    dict(B=3, Ci=32, Co=40, H=17, W=16, k=3, stride=1, padding=0, dilation=1, groups=1),
    dict(B=2, Ci=64, Co=64, H=12, W=13, k=3, stride=1, padding=2, dilation=2, groups=2),
    dict(B=2, Ci=32, Co=32, H=15, W=14, k=(3, 1), stride=2, padding=(1, 0), dilation=1, groups=1),
    dict(B=1, Ci=64, Co=96, H=9, W=9, k=1, stride=1, padding=0, dilation=1, groups=1),
    # shifted-row kernel (conv_nhwc.hip) for forward AND dgrad: Ci % 32 == Co % 32 == 0
    dict(B=3, Ci=32, Co=160, H=23, W=19, k=3, stride=1, padding=(2, 1), dilation=(2, 1), groups=1),
    dict(B=2, Ci=64, Co=32, H=14, W=37, k=(3, 2), stride=1, padding=(1, 0), dilation=1, groups=1),
    dict(B=5, Ci=96, Co=64, H=11, W=10, k=(1, 5), stride=1, padding=(0, 4), dilation=1, groups=1),
])
def test_conv_bf16_fast_path_vs_oracle(cfg):
    """Shapes that take the bf16-MFMA conv kernels (K % 32 == 0): forward, dgrad (incl. the
    generic fallback for stride 2), split-K wgrad, bias; oracle on the bf16-rounded operands."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import Cplx, cplx
    rs = np.random.RandomState(cfg["Ci"] + cfg["Co"])
    B, Ci, Co, H, W, g = cfg["B"], cfg["Ci"], cfg["Co"], cfg["H"], cfg["W"], cfg["groups"]
    kh, kw = (cfg["k"], cfg["k"]) if isinstance(cfg["k"], int) else cfg["k"]
    xr, xi = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(B, Ci, H, W) + 0.2)
    wr, wi = bf16_round(rs.randn(Co, Ci // g, kh, kw) * 0.1), bf16_round(rs.randn(Co, Ci // g, kh, kw) * 0.1)
    br, bi = rs.randn(Co).astype(np.float32), rs.randn(Co).astype(np.float32)
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    txr, txi = q(xr).requires_grad_(True), q(xi).requires_grad_(True)
    twr, twi = T(wr).requires_grad_(True), T(wi).requires_grad_(True)
    tbr, tbi = T(br).requires_grad_(True), T(bi).requires_grad_(True)
    kw_ = dict(stride=cfg["stride"], padding=cfg["padding"], dilation=cfg["dilation"], groups=g)
    y = cplx.conv2d(Cplx(txr, txi), Cplx(twr, twi), Cplx(tbr, tbi), **kw_)
    f = np.float64
    yr, yi = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br.astype(f),
                             bi.astype(f), **kw_)
    assert tuple(y.shape) == yr.shape
    np.testing.assert_allclose(N(y.real), yr, rtol=1e-2, atol=1e-2 * np.abs(yr).max())
    np.testing.assert_allclose(N(y.imag), yi, rtol=1e-2, atol=1e-2 * np.abs(yi).max())
    gr, gi = bf16_round(rs.randn(*yr.shape)), bf16_round(rs.randn(*yr.shape))
    ((y.real * q(gr)).sum() + (y.imag * q(gi)).sum()).backward()
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f),
                             wi.astype(f), **kw_)
    got = dict(dxr=txr.grad, dxi=txi.grad, dwr=twr.grad, dwi=twi.grad, dbr=tbr.grad, dbi=tbi.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), bw[n], rtol=2e-2, atol=2e-2 * np.abs(bw[n]).max(), err_msg=n)


@pytest.mark.parametrize("B,C,H,W,ph,pw", [(2, 64, 9, 13, 0, 0), (3, 40, 7, 70, 2, 1), (1, 136, 5, 4, 1, 3)])
def test_nhwc_pad_exact(B, C, H, W, ph, pw):
    """cplxamd_nhwc_pad is a pure data movement: identical to permute + zero pad."""
    from gpu_util import DEV
    from cplxmodule_amd import conv
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=DEV).bfloat16()
    got = conv.nhwc_pad(x, ph, pw)
    ref = torch.nn.functional.pad(x.float(), (pw, pw, ph, ph)).permute(0, 2, 3, 1).contiguous()
    assert got.shape == ref.shape
    assert torch.equal(got.float(), ref)


def test_real_conv_bf16_rows_kernel_vs_oracle():
    """Real convolution (and its dgrad) through the shifted-row kernel."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import conv
    rs = np.random.RandomState(11)
    B, Ci, Co, H, W = 2, 32, 64, 16, 18
    x, w = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(Co, Ci, 3, 3) * 0.1)
    b = rs.randn(Co).astype(np.float32)
    tx, tw, tb = T(x, torch.bfloat16).requires_grad_(True), T(w).requires_grad_(True), T(b).requires_grad_(True)
    y = conv.RealConv2dFn.apply(tx, tw, tb, 1, 1, 1, 1)
    f = np.float64
    ref = orc.real_conv2d(x.astype(f), w.astype(f), padding=1) + b.astype(f)[None, :, None, None]
    np.testing.assert_allclose(N(y), ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max())
    g = bf16_round(rs.randn(*ref.shape))
    (y * T(g, torch.bfloat16)).sum().backward()
    dx, dw = orc.real_conv2d_bwd(g.astype(f), x.astype(f), w.astype(f), padding=1)
    for n, t, r in (("dx", tx.grad, dx), ("dw", tw.grad, dw), ("db", tb.grad, g.astype(f).sum((0, 2, 3)))):
        np.testing.assert_allclose(N(t), r, rtol=2e-2, atol=2e-2 * np.abs(r).max(), err_msg=n)


@pytest.mark.parametrize("seed", range(12))
def test_conv_bf16_rows_kernels_random_shapes(seed):
    """Property test of the channels-last kernels (forward, data gradient, weight gradient, bias
    gradient) on random stride-1 / groups-1 geometries against the exact float32 kernels run on
    the same bf16-rounded operands: odd image sizes, rectangular / dilated kernels, asymmetric
    padding, channel counts that are not multiples of the 64-wide tiles, rows % 256 != 0."""
    from gpu_util import DEV
    from cplxmodule_amd import Cplx, cplx
    rs = np.random.RandomState(100 + seed)
    B = int(rs.randint(1, 5))
    Ci, Co = int(rs.choice([32, 64, 96])), int(rs.choice([32, 40, 64, 72, 160]))
    kh, kw = int(rs.randint(1, 4)), int(rs.randint(1, 5))
    dh, dw = int(rs.randint(1, 3)), int(rs.randint(1, 3))
    ph, pw = int(rs.randint(0, 3)), int(rs.randint(0, 3))
    H = int(rs.randint((kh - 1) * dh + 1, 30)) + 2
    W = int(rs.randint((kw - 1) * dw + 1, 40)) + 2
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16().to(DEV)  # noqa: E731
    xr, xi = mk(B, Ci, H, W), mk(B, Ci, H, W)
    wr, wi = (mk(Co, Ci, kh, kw).float() * 0.1).bfloat16().float(), (mk(Co, Ci, kh, kw).float() * 0.1).bfloat16().float()
    br, bi = mk(Co).float(), mk(Co).float()
    kw_ = dict(stride=1, padding=(ph, pw), dilation=(dh, dw))
    outs = []
    for dt in (torch.bfloat16, torch.float32):
        leaves = [t.to(dt).clone().requires_grad_(True) for t in (xr, xi)] + \
                 [t.clone().requires_grad_(True) for t in (wr, wi, br, bi)]
        y = cplx.conv2d(Cplx(leaves[0], leaves[1]), Cplx(leaves[2], leaves[3]), Cplx(leaves[4], leaves[5]), **kw_)
        gg = torch.Generator(device="cpu").manual_seed(1000 + seed)
        gr = torch.randn(y.real.shape, generator=gg).bfloat16().to(DEV)
        gi = torch.randn(y.real.shape, generator=gg).bfloat16().to(DEV)
        torch.autograd.backward((y.real, y.imag), (gr.to(dt), gi.to(dt)))
        outs.append([y.real.float(), y.imag.float()] + [t.grad.float() for t in leaves])
    names = ["yr", "yi", "dxr", "dxi", "dwr", "dwi", "dbr", "dbi"]
    for n, a, b in zip(names, *outs):
        scale = float(b.detach().abs().max()) + 1e-6
        tol = 2e-2 if n[0] in "yd" and n[1] in "rix" else 1e-3   # bf16-rounded outputs vs fp32-accumulated grads
        assert float((a.detach() - b.detach()).abs().max()) <= tol * scale, (n, seed, B, Ci, Co, kh, kw, dh, dw, ph, pw, H, W)


C1_CASES = {"s2p1": dict(stride=2, padding=1), "d2": dict(dilation=2),
            "circ": dict(padding=3, padding_mode="circular"), "g2": dict(groups=2, padding=2)}


@pytest.mark.parametrize("name", list(C1_CASES))
def test_conv1d_golden(golden, name):
    """cplx.conv1d (the 2-d kernels on a height-1 image) against the reference: values and every
    gradient, zeros / circular padding, stride, dilation, groups."""
    from gpu_util import T, N
    from cplxmodule_amd import Cplx, cplx
    g = golden("extras")
    k = f"f32_c1_{name}_"
    lv = {n: T(g[k + n]).requires_grad_(True) for n in ("xr", "xi", "wr", "wi", "br", "bi")}
    y = cplx.conv1d(Cplx(lv["xr"], lv["xi"]), Cplx(lv["wr"], lv["wi"]), Cplx(lv["br"], lv["bi"]), **C1_CASES[name])
    np.testing.assert_allclose(N(y.real), g[k + "yr"], **_tol(g[k + "yr"], 2e-5))
    np.testing.assert_allclose(N(y.imag), g[k + "yi"], **_tol(g[k + "yi"], 2e-5))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    for n in ("xr", "xi", "wr", "wi", "br", "bi"):
        np.testing.assert_allclose(N(lv[n].grad), g[k + "d" + n], **_tol(g[k + "d" + n], 5e-5), err_msg=n)


def test_conv1d_layers_and_lrt():
    """CplxConv1d / CplxConv1dVD / Conv1dARD: state-dict layout, eval = mean path, training = the
    2-d LRT on the lifted tensors (same noise position => same bits), penalties, masks."""
    from gpu_util import DEV
    from cplxmodule_amd import Cplx, nn
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(3)
    l1 = rel.CplxConv1dVD(8, 12, 5, stride=2, padding=2).to(DEV)
    l2 = rel.CplxConv2dVD(8, 12, (1, 5), stride=(1, 2), padding=(0, 2)).to(DEV)
    with torch.no_grad():
        for a, b in ((l1.weight.real, l2.weight.real), (l1.weight.imag, l2.weight.imag), (l1.log_sigma2, l2.log_sigma2)):
            a.uniform_(-0.3, 0.3)
            b.copy_(a.unsqueeze(2))
        l1.log_sigma2.sub_(4.0); l2.log_sigma2.sub_(4.0)
        l2.bias.real.copy_(l1.bias.real); l2.bias.imag.copy_(l1.bias.imag)
    assert list(l1.state_dict()) == ["log_sigma2", "weight.imag", "weight.real", "bias.imag", "bias.real"]
    x = Cplx(torch.randn(4, 8, 33, device=DEV, requires_grad=True), torch.randn(4, 8, 33, device=DEV))
    x2 = Cplx(x.real.detach().unsqueeze(2).requires_grad_(True), x.imag.unsqueeze(2))
    for mode in ("eval", "train"):
        getattr(l1, mode)(); getattr(l2, mode)()
        rel.noise.manual_seed(4)
        y1 = l1(x)
        rel.noise.manual_seed(4)
        y2 = l2(x2)
        assert y1.real.shape == (4, 12, 17)
        assert torch.equal(y1.real, y2.real.squeeze(2)) and torch.equal(y1.imag, y2.imag.squeeze(2))
    (y1.real.sum() + sum(rel.penalties(l1))).backward()
    (y2.real.sum() + sum(rel.penalties(l2))).backward()
    assert torch.allclose(l1.log_sigma2.grad, l2.log_sigma2.grad.squeeze(2), rtol=1e-6, atol=1e-7)
    assert torch.allclose(l1.weight.real.grad, l2.weight.real.grad.squeeze(2), rtol=1e-6, atol=1e-7)
    assert torch.allclose(x.real.grad, x2.real.grad.squeeze(2), rtol=1e-6, atol=1e-7)
    assert rel.compute_ard_masks(l1, hard=True, threshold=0.0)["mask"].shape == (12, 8, 5)
    r = rel.Conv1dARD(8, 12, 3, padding=1).to(DEV)
    xr = torch.randn(4, 8, 20, device=DEV)
    assert r(xr).shape == (4, 12, 20)
    r.eval()
    assert torch.allclose(r(xr), torch.nn.functional.conv1d(xr, r.weight, r.bias, padding=1), rtol=1e-4, atol=1e-5)
    c = nn.CplxConv1d(8, 12, 3, padding=2, padding_mode="circular").to(DEV)
    assert c(Cplx(xr, xr)).real.shape == (4, 12, 20)


@pytest.mark.parametrize("cfg", [
    dict(B=2, Ci=32, Co=64, H=14, W=17, k=3, padding=1, dilation=1),
    dict(B=3, Ci=16, Co=40, H=12, W=11, k=(3, 2), padding=(2, 0), dilation=(2, 1)),
    dict(B=1, Ci=48, Co=16, H=9, W=20, k=(1, 4), padding=(0, 3), dilation=1),
    dict(B=2, Ci=64, Co=96, H=10, W=10, k=1, padding=0, dilation=1),
])
def test_conv_f32_rows_kernel_vs_oracle(cfg):
    """The exact-float32 shifted-row kernel (conv_nhwc_f32.hip: forward and data gradient) against
    the float64 oracle at float32 tolerances (1e-5 relative to the largest output)."""
    from gpu_util import T, N
    from cplxmodule_amd import Cplx, cplx
    rs = np.random.RandomState(cfg["Ci"] * 3 + cfg["Co"])
    B, Ci, Co, H, W = cfg["B"], cfg["Ci"], cfg["Co"], cfg["H"], cfg["W"]
    kh, kw = (cfg["k"], cfg["k"]) if isinstance(cfg["k"], int) else cfg["k"]
    f32 = np.float32
    xr, xi = rs.randn(B, Ci, H, W).astype(f32), rs.randn(B, Ci, H, W).astype(f32)
    wr, wi = (rs.randn(Co, Ci, kh, kw) * 0.1).astype(f32), (rs.randn(Co, Ci, kh, kw) * 0.1).astype(f32)
    br, bi = rs.randn(Co).astype(f32), rs.randn(Co).astype(f32)
    lv = [T(a).requires_grad_(True) for a in (xr, xi, wr, wi, br, bi)]
    kw_ = dict(stride=1, padding=cfg["padding"], dilation=cfg["dilation"])
    y = cplx.conv2d(Cplx(lv[0], lv[1]), Cplx(lv[2], lv[3]), Cplx(lv[4], lv[5]), **kw_)
    f = np.float64
    yr, yi = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br.astype(f), bi.astype(f), **kw_)
    s = max(np.abs(yr).max(), np.abs(yi).max())
    np.testing.assert_allclose(N(y.real), yr, rtol=1e-5, atol=1e-5 * s)
    np.testing.assert_allclose(N(y.imag), yi, rtol=1e-5, atol=1e-5 * s)
    gr, gi = rs.randn(*yr.shape).astype(f32), rs.randn(*yr.shape).astype(f32)
    torch.autograd.backward((y.real, y.imag), (T(gr), T(gi)))
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), **kw_)
    for n, t in dict(dxr=lv[0].grad, dxi=lv[1].grad, dwr=lv[2].grad, dwi=lv[3].grad).items():
        np.testing.assert_allclose(N(t), bw[n], rtol=1e-5, atol=2e-5 * np.abs(bw[n]).max(), err_msg=n)
