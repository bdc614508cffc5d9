"""float64 models (cplxmodule_amd/f64.py, csrc/f64.hip): the f64 halves of the reference's golden vectors -- linear, matmul,
LRT layers, penalties / Ei / masks, convolutions (every geometry of the conv fixture, LRT conv layers), batch-norm -- at
1e-10 (the reference's own module tests run `.double()` models: /root/reference/tests/test_modules.py:88-127)."""
import numpy as np
import pytest
import torch

from oracle.gen_golden_cases import CONV_CASES

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def N(t):
    return t.detach().cpu().numpy()


def close(got, ref, what="", r=1e-10):
    assert got.dtype == torch.float64, (what, got.dtype)
    np.testing.assert_allclose(N(got), ref, rtol=r, atol=r * 0.1 * float(np.abs(ref).max()) + 1e-300, err_msg=what)


@pytest.mark.parametrize("case", "abc")
def test_cplx_linear_f64_golden(golden, case):
    from cplxmodule_amd import cplx
    g = golden("linear")
    k = f"f64_{case}_"
    L = {n: T(g[k + n]).requires_grad_(True) for n in ("xr", "xi", "wr", "wi", "br", "bi")}
    x, w, b = cplx.Cplx(L["xr"], L["xi"]), cplx.Cplx(L["wr"], L["wi"]), cplx.Cplx(L["br"], L["bi"])
    y = cplx.linear(x, w, b)
    close(y.real, g[k + "y_naive_r"]); close(y.imag, g[k + "y_naive_i"])
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    for n, m in (("xr", "dxr"), ("xi", "dxi"), ("wr", "dwr"), ("wi", "dwi"), ("br", "dbr"), ("bi", "dbi")):
        close(L[n].grad, g[k + m], m)
    for algo in ("3m", "cat"):
        y = getattr(cplx, "linear_" + algo)(x, w, b)
        close(y.real, g[k + f"y_{algo}_r"]); close(y.imag, g[k + f"y_{algo}_i"])
    close(cplx.linear(x, w, None).real, g[k + "y_nobias_r"])


def test_matmul_f64_golden_and_layer(golden):
    from cplxmodule_amd import cplx, nn
    g = golden("linear")
    k = "f64_mm_"
    u, v = cplx.Cplx(T(g[k + "ur"]), T(g[k + "ui"])), cplx.Cplx(T(g[k + "vr"]), T(g[k + "vi"]))
    m = u @ v
    close(m.real, g[k + "mr"]); close(m.imag, g[k + "mi"])
    close((u[0] @ v[0]).imag, g[k + "mi"][0])
    layer = nn.CplxLinear(200, 321).to("cuda").double()          # the module in .double(), as the reference's tests build it
    k = "f64_a_"
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
                           "bias.imag": T(g[k + "bi"])})
    y = layer(cplx.Cplx(T(g[k + "xr"]), T(g[k + "xi"])))
    close(y.real, g[k + "y_naive_r"])
    assert abs(y).dtype == torch.float64


def test_lrt_linear_f64_goldens(golden):
    from cplxmodule_amd import cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("lrt_linear")
    k = "f64_cplx_"
    layer = rel.CplxLinearVD(128, 128).to("cuda").double()
    sd = {"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
          "bias.imag": T(g[k + "bi"]), "log_sigma2": T(g[k + "ls2"])}
    layer.load_state_dict(sd)
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    tape = T(g[k + "tape"]) / np.sqrt(2.0)
    layer.train()
    y = layer(cplx.Cplx(xr, xi), eps=cplx.Cplx(tape[0], tape[1]))
    close(y.real, g[k + "yr"]); close(y.imag, g[k + "yi"])
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad,
               dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        close(t, g[k + n], n)
    layer.eval()
    close(layer(cplx.Cplx(xr, xi)).real, g[k + "yr_eval"])
    # clamp boundary
    sd["log_sigma2"] = T(g[k + "clamp_ls2"])
    layer.load_state_dict(sd)
    layer.train(); layer.zero_grad()
    xr, xi = T(g[k + "clamp_xr"]).requires_grad_(True), T(g[k + "clamp_xi"]).requires_grad_(True)
    tape = T(g[k + "clamp_tape"]) / np.sqrt(2.0)
    y = layer(cplx.Cplx(xr, xi), eps=cplx.Cplx(tape[0], tape[1]))
    close(y.real, g[k + "clamp_yr"])
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    close(layer.log_sigma2.grad, g[k + "clamp_dls2"]); close(xr.grad, g[k + "clamp_dxr"])
    # real layer
    k = "f64_real_"
    layer = rel.LinearVD(128, 128).to("cuda").double()
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x = T(g[k + "x"]).requires_grad_(True)
    layer.train()
    y = layer(x, eps=T(g[k + "eps"]))
    close(y, g[k + "y"])
    (y * T(g[k + "g"])).sum().backward()
    for n, t in dict(dx=x.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad).items():
        close(t, g[k + n], n)
    # without a given noise tensor the float64 layer draws torch's: finite, right shape, different per call
    y1, y2 = layer(x), layer(x)
    assert y1.dtype == torch.float64 and not torch.equal(y1, y2)


@pytest.mark.parametrize("kind,cls", [("cplx_vd", "CplxLinearVD"), ("cplx_ard", "CplxLinearARD"),
                                      ("real_vd", "LinearVD"), ("real_ard", "LinearARD")])
def test_penalty_masks_f64_golden(golden, kind, cls):
    from cplxmodule_amd.nn import relevance as rel
    g = golden("penalty")
    O, I = g["f64_ls2"].shape
    layer = getattr(rel, cls)(I, O, bias=False).to("cuda").double()
    if kind.startswith("cplx"):
        layer.load_state_dict({"weight.real": T(g["f64_wr"]), "weight.imag": T(g["f64_wi"]), "log_sigma2": T(g["f64_ls2"])})
        params = [layer.log_sigma2, layer.weight.real, layer.weight.imag]
        names = ["dls2", "dwr", "dwi"]
    else:
        layer.load_state_dict({"weight": T(g["f64_wr"]), "log_sigma2": T(g["f64_ls2"])})
        params = [layer.log_sigma2, layer.weight]
        names = ["dls2", "dwr"]
    fin = np.isfinite(g[f"f64_{kind}_penalty"])
    close(layer.log_alpha, g[f"f64_{kind}_log_alpha"], "log_alpha")
    pen = layer.penalty
    np.testing.assert_allclose(N(pen)[fin], g[f"f64_{kind}_penalty"][fin], rtol=1e-10, atol=1e-12)
    grads = torch.autograd.grad((pen * T(g["f64_g"]))[torch.from_numpy(fin).cuda()].sum(), params)
    for n, t in zip(names, grads):
        ref = g[f"f64_{kind}_{n}"]
        ok = np.isfinite(ref)
        np.testing.assert_allclose(N(t)[ok], ref[ok], rtol=1e-9, atol=1e-11 * float(np.abs(ref[ok]).max()), err_msg=n)
    if np.isfinite(g[f"f64_{kind}_sum"]):
        tot = sum(rel.penalties(layer, reduction="sum"))
        np.testing.assert_allclose(float(tot), float(g[f"f64_{kind}_sum"]), rtol=1e-11)
        np.testing.assert_allclose(float(sum(rel.penalties(layer, reduction="mean"))), float(g[f"f64_{kind}_mean"]), rtol=1e-11)
    for th in (-0.5, 1.0, 3.0):
        m = layer.relevance(threshold=th)
        assert m.dtype == torch.float64
        np.testing.assert_array_equal(N(m), g[f"f64_{kind}_mask_{th}"])
    assert isinstance(layer.sparsity(threshold=1.0), list)


def test_expi_f64_golden(golden):
    from cplxmodule_amd.nn.relevance.complex import torch_expi
    g = golden("penalty")
    x = T(g["f64_expi_x"]).requires_grad_(True)
    y = torch_expi(x)
    ref = g["f64_expi_y"]
    ok = np.isfinite(ref)
    np.testing.assert_allclose(N(y)[ok], ref[ok], rtol=1e-12, atol=1e-300)
    y[torch.from_numpy(ok).cuda()].sum().backward()
    np.testing.assert_allclose(N(x.grad)[ok], g["f64_expi_dx"][ok], rtol=1e-12)


@pytest.mark.parametrize("case", list(CONV_CASES))
def test_cplx_conv2d_f64_golden(golden, case):
    from cplxmodule_amd import Cplx, nn
    g = golden("conv")
    B, Ci, Co, H, W, ks, st, pd, dl, gp, mode = CONV_CASES[case]
    k = f"f64_{case}_"
    layer = nn.CplxConv2d(Ci, Co, ks, stride=st, padding=pd, dilation=dl, groups=gp, padding_mode=mode).to("cuda").double()
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
                           "bias.imag": T(g[k + "bi"])})
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    y = layer(Cplx(xr, xi))
    close(y.real, g[k + "yr"]); close(y.imag, g[k + "yi"])
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad, dbr=layer.bias.real.grad,
               dbi=layer.bias.imag.grad)
    for n, t in got.items():
        close(t, g[k + n], n)


def test_lrt_conv_f64_goldens(golden):
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    g = golden("conv")
    k = "f64_lrtc_"
    layer = rel.CplxConv2dVD(3, 4, 3, stride=1, padding=1).to("cuda").double()
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
                           "bias.imag": T(g[k + "bi"]), "log_sigma2": T(g[k + "ls2"])})
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    tape = T(g[k + "tape"]) / np.sqrt(2.0)
    layer.train()
    y = layer(Cplx(xr, xi), eps=Cplx(tape[0], tape[1]))
    close(y.real, g[k + "yr"]); close(y.imag, g[k + "yi"])
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum()).backward()
    got = dict(dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad, dbr=layer.bias.real.grad,
               dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    for n, t in got.items():
        close(t, g[k + n], n)
    np.testing.assert_allclose(float(sum(rel.penalties(layer))), float(g[k + "penalty_sum"]), rtol=1e-11)
    k = "f64_lrtr_"
    layer = rel.Conv2dVD(3, 4, 3, stride=2, padding=1).to("cuda").double()
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x = T(g[k + "x"]).requires_grad_(True)
    layer.train()
    y = layer(x, eps=T(g[k + "eps"]))
    close(y, g[k + "y"])
    (y * T(g[k + "g"])).sum().backward()
    for n, t in dict(dx=x.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad).items():
        close(t, g[k + n], n)


@pytest.mark.parametrize("name,cls", [("2d", "CplxBatchNorm2d"), ("1d", "CplxBatchNorm1d"), ("1d3", "CplxBatchNorm1d")])
def test_batchnorm_f64_golden(golden, name, cls):
    from cplxmodule_amd import Cplx, nn
    g = golden("batchnorm")
    k = f"f64_{name}_"
    F_ = g[k + "weight"].shape[-1]
    bn = getattr(nn, cls)(F_).to("cuda").double()
    with torch.no_grad():
        bn.weight.copy_(T(g[k + "weight"])); bn.bias.copy_(T(g[k + "bias"]))
    bn.train()
    for step in range(3):
        s = k + f"s{step}_"
        xr, xi = T(g[s + "xr"]).requires_grad_(True), T(g[s + "xi"]).requires_grad_(True)
        y = bn(Cplx(xr, xi))
        close(y.real, g[s + "yr"]); close(y.imag, g[s + "yi"])
        close(bn.running_mean, g[s + "running_mean"]); close(bn.running_var, g[s + "running_var"])
        assert int(bn.num_batches_tracked) == int(g[s + "nbt"])
        bn.zero_grad()
        ((y.real * T(g[s + "gr"])).sum() + (y.imag * T(g[s + "gi"])).sum()).backward()
        for n, t in dict(dxr=xr.grad, dxi=xi.grad, dweight=bn.weight.grad, dbias=bn.bias.grad).items():
            close(t, g[s + n], f"{n} step {step}", r=1e-9)
    bn.eval(); bn.zero_grad()
    xr, xi = xr.detach().requires_grad_(True), xi.detach().requires_grad_(True)
    y = bn(Cplx(xr, xi))
    close(y.real, g[k + "eval_yr"])
    ((y.real * T(g[s + "gr"])).sum() + (y.imag * T(g[s + "gi"])).sum()).backward()
    for n, t in dict(dxr=xr.grad, dxi=xi.grad, dweight=bn.weight.grad, dbias=bn.bias.grad).items():
        close(t, g[k + "eval_" + n], n, r=1e-9)


def test_f64_second_derivatives_and_unsupported_layers():
    """Gradient penalty through a float64 CplxLinear + batch-norm (differentiable torch algebra around the library's GEMM,
    whose backward is the GEMM itself) against float64 autograd of the reference's formulas on the CPU; layers without a
    float64 form refuse loudly."""
    from cplxmodule_amd import Cplx, cplx, nn
    from cplxmodule_amd._lib import CplxAmdError
    torch.manual_seed(0)
    base = [torch.randn(12, 10, dtype=torch.float64), torch.randn(12, 10, dtype=torch.float64),
            torch.randn(7, 10, dtype=torch.float64), torch.randn(7, 10, dtype=torch.float64)]

    def run(dev):
        L = [t.to(dev).requires_grad_(True) for t in base]
        if dev == "cpu":
            yr = L[0] @ L[2].t() - L[1] @ L[3].t()
            yi = L[0] @ L[3].t() + L[1] @ L[2].t()
        else:
            y = cplx.linear(Cplx(L[0], L[1]), Cplx(L[2], L[3]))
            yr, yi = y.real, y.imag
        g = torch.autograd.grad((yr ** 3).sum() + (yr * yi).sum(), L, create_graph=True)
        return [t.detach().cpu().numpy() for t in torch.autograd.grad(sum((t ** 2).sum() for t in g), L)]

    for a, b in zip(run("cuda"), run("cpu")):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-11 * float(np.abs(b).max()))
    x = Cplx(torch.randn(4, 6, device="cuda", dtype=torch.float64), torch.randn(4, 6, device="cuda", dtype=torch.float64))
    with pytest.raises(CplxAmdError, match="float64"):
        cplx.modrelu(x, 0.1)
    with pytest.raises(CplxAmdError):
        cplx.linear(x, Cplx(torch.randn(3, 6, device="cuda"), torch.randn(3, 6, device="cuda")))     # mixed precision
    del nn


def test_f64_conv_random_geometries_against_aten():
    """Thirty random (stride, padding, dilation, groups, kernel, channels) draws: the float64 convolution kernel (forward, data
    gradient, weight gradient) against aten's float64 convolution of the same operands -- a checker only."""
    from cplxmodule_amd import Cplx, cplx
    F_ = torch.nn.functional
    rs = np.random.RandomState(5)
    for it in range(30):
        groups = int(rs.choice([1, 1, 2, 3]))
        Ci, Co = groups * int(rs.randint(1, 5)), groups * int(rs.randint(1, 5))
        kh, kw = int(rs.randint(1, 4)), int(rs.randint(1, 4))
        st = (int(rs.randint(1, 3)), int(rs.randint(1, 3)))
        dl = (int(rs.randint(1, 3)), int(rs.randint(1, 3)))
        pd = (int(rs.randint(0, 3)), int(rs.randint(0, 3)))
        B, H, W = int(rs.randint(1, 4)), int(rs.randint(6, 14)), int(rs.randint(6, 14))
        if (H + 2 * pd[0] - dl[0] * (kh - 1) - 1) < 0 or (W + 2 * pd[1] - dl[1] * (kw - 1) - 1) < 0:
            continue
        mk = lambda *s: torch.from_numpy(rs.randn(*s)).cuda().requires_grad_(True)  # noqa: E731
        xr, xi, wr, wi, br, bi = mk(B, Ci, H, W), mk(B, Ci, H, W), mk(Co, Ci // groups, kh, kw), mk(Co, Ci // groups, kh, kw), mk(Co), mk(Co)
        kw_ = dict(stride=st, padding=pd, dilation=dl, groups=groups)
        y = cplx.conv2d(Cplx(xr, xi), Cplx(wr, wi), Cplx(br, bi), **kw_)
        c = lambda a, b: F_.conv2d(a, b, None, **kw_)  # noqa: E731
        ref_r = c(xr, wr) - c(xi, wi) + br.view(1, -1, 1, 1)
        ref_i = c(xr, wi) + c(xi, wr) + bi.view(1, -1, 1, 1)
        assert y.real.dtype == torch.float64 and y.real.shape == ref_r.shape, (it, y.real.shape, ref_r.shape)
        gr, gi = torch.from_numpy(rs.randn(*ref_r.shape)).cuda(), torch.from_numpy(rs.randn(*ref_r.shape)).cuda()
        leaves = [xr, xi, wr, wi, br, bi]
        got = torch.autograd.grad((y.real * gr).sum() + (y.imag * gi).sum(), leaves)
        ref = torch.autograd.grad((ref_r * gr).sum() + (ref_i * gi).sum(), leaves)
        for a, b in [(y.real, ref_r), (y.imag, ref_i)] + list(zip(got, ref)):
            np.testing.assert_allclose(N(a), N(b), rtol=1e-10, atol=1e-11 * float(b.detach().abs().max()), err_msg=str((it, kw_, kh, kw)))
