"""Round-2 GPU parity tests: abs(Cplx) / log_alpha as kernels with their backward (exact zeros),
signed-cotangent KL gradients, the masked layers against reference fixtures, the fused
operand-preparation + KL path of the LRT layers, the extended GEMM epilogue (beta / exp / both planes).
Everything goes through libcplxamd.so (C ABI via ctypes)."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc

pytestmark = pytest.mark.gpu

KINDS = orc.KINDS


def _close(got, ref, rtol=1e-5, atol_rel=1e-5, what=""):
    ref = np.asarray(ref)
    scale = float(np.abs(ref[np.isfinite(ref)]).max()) if np.isfinite(ref).any() else 1.0
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_rel * max(scale, 1e-30), err_msg=what)


# ------------------------------------------------------------------------------------------------
def test_abs_forward_backward_golden(golden):
    """cplxamd_cplx_abs_fwd / _bwd against the reference's stack + norm: value bit-for-bit (the kernel
    reproduces torch's rounding chain), gradient 0 at exact zeros (sqrt chain gave NaN)."""
    from gpu_util import T, N
    from cplxmodule_amd import Cplx
    g = golden("r02")
    zr, zi = T(g["f32_abs_zr"]).requires_grad_(True), T(g["f32_abs_zi"]).requires_grad_(True)
    a = abs(Cplx(zr, zi))
    np.testing.assert_array_equal(N(a), g["f32_abs_abs"])
    (a * T(g["f32_abs_g"])).sum().backward()
    assert torch.isfinite(zr.grad).all() and torch.isfinite(zi.grad).all()
    np.testing.assert_allclose(N(zr.grad), g["f32_abs_dzr"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(N(zi.grad), g["f32_abs_dzi"], rtol=1e-6, atol=1e-7)
    zero = (g["f32_abs_zr"] == 0) & (g["f32_abs_zi"] == 0)
    assert zero.sum() >= 5 and np.all(N(zr.grad)[zero] == 0) and np.all(N(zi.grad)[zero] == 0)
    # bf16 planes: value within one bf16 ulp of the float32 result, finite gradients at zeros
    zb, zc = T(g["f32_abs_zr"]).bfloat16().requires_grad_(True), T(g["f32_abs_zi"]).bfloat16().requires_grad_(True)
    ab = abs(Cplx(zb, zc))
    ref = np.sqrt(N(zb) ** 2 + N(zc) ** 2)
    np.testing.assert_allclose(N(ab), ref, rtol=2 ** -8, atol=1e-30)
    ab.float().sum().backward()
    assert torch.isfinite(zb.grad.float()).all()


@pytest.mark.parametrize("kind", KINDS)
def test_log_alpha_differentiable_golden(golden, kind):
    from gpu_util import T, N
    from cplxmodule_amd.nn import relevance as rel
    g = golden("r02")
    cls = {"real_vd": rel.LinearVD, "real_ard": rel.LinearARD, "cplx_vd": rel.CplxLinearVD,
           "cplx_ard": rel.CplxLinearARD}[kind]
    O, I = g["f32_sg_wr"].shape
    layer = cls(I, O, bias=False).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.copy_(T(g["f32_sg_ls2"]))
        if kind.startswith("cplx"):
            layer.weight.real.copy_(T(g["f32_sg_wr"]))
            layer.weight.imag.copy_(T(g["f32_sg_wi"]))
            wps = [layer.weight.real, layer.weight.imag]
        else:
            layer.weight.copy_(T(g["f32_sg_wr"]))
            wps = [layer.weight]
    k = f"f32_sg_{kind}_"
    la = layer.log_alpha
    assert la.requires_grad
    np.testing.assert_array_equal(N(la), g[k + "la"])            # exact-log kernel == reference bits
    grads = torch.autograd.grad((la * T(g["f32_sg_g"])).sum(), [layer.log_sigma2] + wps)
    np.testing.assert_array_equal(N(grads[0]), g["f32_sg_g"])
    for got, name in zip(grads[1:], ("la_dwr", "la_dwi")):
        ref = g[k + name]
        assert np.isfinite(N(got)).all()
        np.testing.assert_allclose(N(got), ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("kind", KINDS)
def test_penalty_gradients_signed_cotangent(golden, kind):
    """Mixed-sign and negative upstream gradients through `.penalty` and through the fused sum
    (copysignf dropped the sign of the upstream gradient for the real kinds)."""
    from gpu_util import T, N
    from cplxmodule_amd import ops
    g = golden("r02")
    wr, ls2, gs = g["f32_sg_wr"], g["f32_sg_ls2"], g["f32_sg_g"]
    wi = g["f32_sg_wi"] if kind.startswith("cplx") else None
    twi = None if wi is None else T(wi)
    k = f"f32_sg_{kind}_"
    f = np.float64
    o = orc.penalty_bwd(kind, gs.astype(f), ls2.astype(f), wr.astype(f), None if wi is None else wi.astype(f))
    got = ops.kl_bwd(kind, T(wr), twi, T(ls2), g_elem=T(gs))
    neg = ops.kl_bwd(kind, T(wr), twi, T(ls2), g_scalar=torch.tensor(-0.37, device="cuda"))
    theta = np.abs(wr) if wi is None else np.sqrt(wr.astype(f) ** 2 + wi.astype(f) ** 2)
    with np.errstate(divide="ignore"):
        amp = np.where(theta > 0, 2 / (theta + 1e-12), 0)
    eps = np.finfo(np.float32).eps
    o_neg = orc.penalty_bwd(kind, np.full_like(gs, -0.37, dtype=f), ls2.astype(f), wr.astype(f),
                            None if wi is None else wi.astype(f))
    for j, (name, okey) in enumerate((("dls2", "dlog_sigma2"), ("dwr", "dwr"), ("dwi", "dwi"))):
        if got[j] is None:
            continue
        a = amp if j else 1.0
        for res, oo, gold in ((got[j], o[okey], g[k + "pen_" + name]), (neg[j], o_neg[okey], g[k + "negsum_" + name])):
            r = N(res).astype(f)
            # the yardstick of tests/test_gpu_vd.py::test_penalty_gradients: the float64 oracle, with the
            # few-ulp(1) error of f' amplified by 2/|w| (for large log-alpha the REFERENCE's own float32
            # chain 1 - exp(-e^t) cancels, so its values are the looser of the two)
            err = np.abs(r - oo)
            upmax = max(1.0, float(np.abs(gs).max()))           # |upstream| scales the absolute error
            bound = (2e-5 * np.abs(oo) + 8 * eps * np.maximum(a, 1.0) * upmax) * np.ones_like(err)
            fin = np.isfinite(oo)
            assert (err[fin] <= bound[fin]).all(), (kind, name, float((err - bound)[fin].max()))
            # sign agreement wherever the gradient is non-zero (the r1 bug: copysignf dropped it)
            nz = fin & (np.abs(oo) > 1e-12)      # (below that float32 intermediates may flush to zero)
            assert np.all(np.sign(r[nz]) == np.sign(oo[nz])), (kind, name)
            # and the reference's own float32 values, within what ITS chain can deliver: f' = 1 - exp(-e^t) is
            # quantised to ulp(1) there, i.e. an absolute error of a few eps amplified by 2/|w|
            m = np.isfinite(gold)
            assert (np.abs(r - gold)[m] <= (1e-4 * np.abs(gold) + 16 * eps * np.maximum(a, 1.0) * upmax *
                                            np.ones_like(err))[m]).all(), (kind, name, "vs reference")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mname", ["hard", "soft"])
def test_cplx_linear_masked_golden(golden, mname):
    """CplxLinearMasked against the REFERENCE's masked layer (not against our own dense layer)."""
    from gpu_util import T, N
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import masked
    g = golden("r02")
    k = "f32_mk_cl_"
    O, I = g[k + "wr"].shape
    lay = masked.CplxLinearMasked(I, O, bias=True).to("cuda")
    with torch.no_grad():
        lay.weight.real.copy_(T(g[k + "wr"])); lay.weight.imag.copy_(T(g[k + "wi"]))
        lay.bias.real.copy_(T(g[k + "br"])); lay.bias.imag.copy_(T(g[k + "bi"]))
    lay.mask = T(g[k + mname + "_mask"])
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    y = lay(Cplx(xr, xi))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    kk = k + mname + "_"
    for got, name in ((y.real, "yr"), (y.imag, "yi"), (xr.grad, "dxr"), (xi.grad, "dxi"),
                      (lay.weight.real.grad, "dwr"), (lay.weight.imag.grad, "dwi"),
                      (lay.bias.real.grad, "dbr"), (lay.bias.imag.grad, "dbi")):
        _close(N(got), g[kk + name], what=name)
    # masked-out weights get an exactly zero gradient (the mask is applied in the GEMM epilogue)
    dropped = g[kk + "mask"] == 0
    assert np.all(N(lay.weight.real.grad)[dropped] == 0) and np.all(N(lay.weight.imag.grad)[dropped] == 0)
    assert sorted(lay.state_dict().keys()) == list(g[k + "state_keys"])
    if mname == "soft":            # the fixture's sparsity numbers were taken with the soft mask in place
        np.testing.assert_allclose([v for _, v in lay.sparsity(hard=True)], g[k + "sparsity_hard"])
        np.testing.assert_allclose([v for _, v in lay.sparsity(hard=False)], g[k + "sparsity_soft"], rtol=1e-6)
    # bf16 activations: the mask rides in the fp32 -> bf16 operand conversion (one kernel)
    yb = lay(Cplx(xr.detach().bfloat16(), xi.detach().bfloat16()))
    _close(N(yb.real), g[kk + "yr"], rtol=3e-2, atol_rel=2e-2)


def test_real_and_conv_masked_golden(golden):
    from gpu_util import T, N
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import masked
    g = golden("r02")
    k = "f32_mk_rl_"
    O, I = g[k + "w"].shape
    rl = masked.LinearMasked(I, O, bias=True).to("cuda")
    with torch.no_grad():
        rl.weight.copy_(T(g[k + "w"])); rl.bias.copy_(T(g[k + "b"]))
    rl.mask = T(g["f32_mk_cl_soft_mask"])
    x = T(g[k + "x"]).requires_grad_(True)
    y = rl(x)
    y.backward(T(g["f32_mk_cl_gr"]))
    for got, name in ((y, "y"), (x.grad, "dx"), (rl.weight.grad, "dw"), (rl.bias.grad, "db")):
        _close(N(got), g[k + name], what=name)
    k = "f32_mk_cc_"
    cl = masked.CplxConv2dMasked(4, 6, 3, padding=1).to("cuda")
    with torch.no_grad():
        cl.weight.real.copy_(T(g[k + "wr"])); cl.weight.imag.copy_(T(g[k + "wi"]))
        cl.bias.real.copy_(T(g[k + "br"])); cl.bias.imag.copy_(T(g[k + "bi"]))
    cl.mask = T(g[k + "mask"])
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    y = cl(Cplx(xr, xi))
    torch.autograd.backward((y.real, y.imag), (T(g[k + "gr"]), T(g[k + "gi"])))
    for got, name in ((y.real, "yr"), (y.imag, "yi"), (xr.grad, "dxr"), (xi.grad, "dxi"),
                      (cl.weight.real.grad, "dwr"), (cl.weight.imag.grad, "dwi"), (cl.bias.real.grad, "dbr")):
        _close(N(got), g[k + name], what=name)
    k = "f32_mk_rc_"
    rc = masked.Conv2dMasked(4, 6, 3, padding=1).to("cuda")
    with torch.no_grad():
        rc.weight.copy_(T(g[k + "w"])); rc.bias.copy_(T(g[k + "b"]))
    rc.mask = T(g["f32_mk_cc_mask"])
    x = T(g[k + "x"]).requires_grad_(True)
    y = rc(x)
    y.backward(T(g["f32_mk_cc_gr"]))
    for got, name in ((y, "y"), (x.grad, "dx"), (rc.weight.grad, "dw"), (rc.bias.grad, "db")):
        _close(N(got), g[k + name], what=name)


def test_mask_mul_kernel():
    from cplxmodule_amd import ops
    torch.manual_seed(3)
    for n in (1, 7, 1024, 4099):
        a, b = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
        m = (torch.rand(n, device="cuda") > 0.5).float() * torch.rand(n, device="cuda")
        r, i = ops.mask_mul(a, b, m)
        assert torch.equal(r, a * m) and torch.equal(i, b * m)
        r, _ = ops.mask_mul(a, None, m, out_dtype=torch.bfloat16)
        assert torch.equal(r, (a * m).bfloat16())
        r, i = ops.mask_mul(a.bfloat16(), b.bfloat16(), m, out_dtype=torch.float32)
        assert torch.equal(r, a.bfloat16().float() * m)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(96, 64, 128), (300, 72, 40), (256, 256, 512)])
def test_gemm_epilogue_extensions(dtype, shape):
    """C = (A^T B) * exp(E) + beta * C (real) and C = (A^T conj B) * M + beta * C on both planes
    (complex) against float64 numpy: the fused-KL backward and the masked weight gradient."""
    from gpu_util import N
    from cplxmodule_amd import ops
    B, O, I = shape
    torch.manual_seed(5)
    g = torch.randn(B, O, device="cuda").to(dtype)
    a = torch.rand(B, I, device="cuda").to(dtype)
    ls2 = torch.empty(O, I, device="cuda").uniform_(-4, 1)
    c0 = torch.randn(O, I, device="cuda")
    beta = torch.tensor(-0.7, device="cuda")
    f = np.float64
    ref = (N(g).astype(f).T @ N(a).astype(f)) * np.exp(N(ls2).astype(f)) + (-0.7) * N(c0).astype(f)
    out = c0.clone()
    ops._real_linear_dw(g, a, emul=ls2, emul_exp=True, out=out, accumulate=True, beta=beta)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    _close(N(out), ref, rtol=tol, atol_rel=tol / 4)
    # plain accumulate (beta None) still adds once
    out2 = c0.clone()
    ops._real_linear_dw(g, a, out=out2, accumulate=True)
    _close(N(out2), N(g).astype(f).T @ N(a).astype(f) + N(c0).astype(f), rtol=tol, atol_rel=tol / 4)
    gi, xi = torch.randn(B, O, device="cuda").to(dtype), torch.randn(B, I, device="cuda").to(dtype)
    xr = torch.randn(B, I, device="cuda").to(dtype)
    mask = (torch.rand(O, I, device="cuda") > 0.5).float()
    cr0, ci0 = torch.randn(O, I, device="cuda"), torch.randn(O, I, device="cuda")
    cr, ci = cr0.clone(), ci0.clone()
    ops._cplx_linear_dw(g, gi, xr, xi, out=(cr, ci), accumulate=True, beta=beta, emul=mask)
    G = N(g).astype(f) + 1j * N(gi).astype(f)
    X = N(xr).astype(f) + 1j * N(xi).astype(f)
    W = (G.T @ X.conj()) * N(mask).astype(f) - 0.7 * (N(cr0).astype(f) + 1j * N(ci0).astype(f))
    _close(N(cr), W.real, rtol=tol, atol_rel=tol / 4)
    _close(N(ci), W.imag, rtol=tol, atol_rel=tol / 4)


def _run_step(layer, x, eps, klw, rel, cplx_):
    """forward + KL + backward; returns (y planes, kl, grads of every parameter and of x)."""
    for p in layer.parameters():
        p.grad = None
    xs = [t.detach().clone().requires_grad_(True) for t in x]
    if cplx_:
        from cplxmodule_amd import Cplx
        y = layer(Cplx(*xs), eps=None if eps is None else Cplx(*eps))
        ys = (y.real, y.imag)
    else:
        y = layer(xs[0], eps=eps)
        ys = (y,)
    kl = sum(rel.penalties(layer, reduction="sum"))
    loss = sum((t.float() ** 2).sum() for t in ys) + klw * kl
    loss.backward()
    return ([t.detach().float().cpu().numpy() for t in ys], float(kl),
            {n: p.grad.detach().float().cpu().numpy().copy() for n, p in layer.named_parameters()},
            [t.grad.detach().float().cpu().numpy() for t in xs])


@pytest.mark.parametrize("kind", ["cplx_vd", "cplx_ard", "real_vd", "real_ard"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bias", [True, False])
def test_fused_kl_matches_standalone(kind, dtype, bias):
    """The training forward that carries its own KL term (one fused prep + KL kernel for bf16, the KL
    gradients added in the weight-gradient GEMM epilogues) must give the values and gradients of the
    stand-alone path (which the golden tests pin to the reference): same layer, same noise, step 1 runs
    stand-alone (and arms the fusion), step 2 runs fused."""
    from cplxmodule_amd.nn import relevance as rel
    from cplxmodule_amd import ops
    cplx_ = kind.startswith("cplx")
    cls = {"real_vd": rel.LinearVD, "real_ard": rel.LinearARD, "cplx_vd": rel.CplxLinearVD,
           "cplx_ard": rel.CplxLinearARD}[kind]
    B, I, O = 96, 64, 128
    torch.manual_seed(7)
    layer = cls(I, O, bias=bias).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-9, 1)
    x = [torch.randn(B, I, device="cuda").to(dtype) for _ in range(2 if cplx_ else 1)]
    eps = [torch.randn(B, O, device="cuda").to(dtype) * 0.7 for _ in range(2 if cplx_ else 1)]
    eps = eps if cplx_ else eps[0]
    layer.train()
    klw = -0.25            # a negative coefficient on purpose
    calls = []
    orig = ops.prep_kl
    ops.prep_kl = lambda *a, **k: (calls.append(a[4]), orig(*a, **k))[1]
    try:
        y1, kl1, g1, dx1 = _run_step(layer, x, eps, klw, rel, cplx_)
        assert not any(calls), "step 1 must not have carried the KL"
        assert layer._kl_fuse
        y2, kl2, g2, dx2 = _run_step(layer, x, eps, klw, rel, cplx_)
        if dtype == torch.bfloat16:
            assert any(calls), "step 2 (bf16) must have run the fused prep + KL kernel"
    finally:
        ops.prep_kl = orig
    assert layer._kl_cache is not None and layer._kl_cache[2][0]
    tol = 2e-6 if dtype == torch.float32 else 1e-5   # same kernels on the data path: near bit-equal
    for a, b in zip(y1, y2):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(kl2, kl1, rtol=1e-6)
    for n in g1:
        _close(g2[n], g1[n], rtol=tol * 10, atol_rel=tol, what=n)
    for a, b in zip(dx1, dx2):
        np.testing.assert_array_equal(a, b)
    # an optimizer-style in-place update invalidates the cached term
    with torch.no_grad():
        layer.log_sigma2.add_(0.01)
    assert layer._kl_get(tuple(layer._kl_cache[1][j][0] for j in range(len(layer._kl_cache[1])))) is None


def test_fused_kl_only_and_retain_graph():
    """Corner cases of the fused term: only the KL reaches the loss; a second backward through a
    retained graph; the KL output unused."""
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(9)
    layer = rel.CplxLinearVD(64, 32).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-8, 0)
    x = Cplx(torch.randn(48, 64, device="cuda").bfloat16(), torch.randn(48, 64, device="cuda").bfloat16())
    layer.train()
    one = lambda: next(iter(rel.penalties(layer)))        # noqa: E731  (sum() would wrap it in an add)
    ref = one()                                           # stand-alone, arms the fusion
    gref = torch.autograd.grad(ref, [layer.log_sigma2, layer.weight.real, layer.weight.imag])
    y = layer(x)                                          # fused now
    kl = one()
    assert kl.grad_fn is not None and type(kl.grad_fn).__name__.startswith("CplxLinearLRTFn")
    np.testing.assert_allclose(float(kl), float(ref), rtol=1e-6)
    g = torch.autograd.grad(kl, [layer.log_sigma2, layer.weight.real, layer.weight.imag], retain_graph=True)
    for a, b in zip(g, gref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
    loss = y.real.float().square().sum() + y.imag.float().square().sum() + 0.5 * kl
    g1 = torch.autograd.grad(loss, [layer.log_sigma2, layer.weight.real], retain_graph=True)
    g2 = torch.autograd.grad(loss, [layer.log_sigma2, layer.weight.real])
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    # KL not used at all: gradients are the data gradients only
    y = layer(x)
    gd = torch.autograd.grad(y.real.float().square().sum() + y.imag.float().square().sum(), [layer.log_sigma2])
    assert torch.isfinite(gd[0]).all()


@pytest.mark.parametrize("rows,cols,dtype", [(20000, 64, torch.bfloat16), (9000, 24, torch.float32), (8192, 512, torch.bfloat16),
                                             (70000, 8, torch.float32), (300, 64, torch.bfloat16), (5000, 4096, torch.bfloat16),
                                             (1000, 20, torch.float32)])
def test_colsum_all_paths(rows, cols, dtype):
    """ops.colsum (bias gradients): the tall-and-narrow row kernel (channels-last activations), the wide partial kernel
    and the scalar fallback against a float64 sum of the same values."""
    from cplxmodule_amd import ops
    torch.manual_seed(rows + cols)
    x = torch.randn(rows, cols, device="cuda").to(dtype)
    got = ops.colsum(x).double().cpu().numpy()
    ref = x.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * float(np.abs(x.double().cpu().numpy()).sum(0).max()))
    out = torch.empty(cols, device="cuda")
    assert ops.colsum(x, out=out) is out
    np.testing.assert_array_equal(out.double().cpu().numpy(), got)
