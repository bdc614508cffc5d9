"""HIP KL / mask / reparameterization kernels against the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc
from oracle import philox

pytestmark = pytest.mark.gpu

KINDS = orc.KINDS


@pytest.fixture(scope="module")
def ops():
    from cplxmodule_amd import ops
    return ops


def _params(g, kind):
    wr, ls2 = g["f32_wr"], g["f32_ls2"]
    wi = g["f32_wi"] if kind.startswith("cplx") else None
    return wr, wi, ls2


@pytest.mark.parametrize("kind", KINDS)
def test_penalty_values_and_sum(golden, ops, kind):
    from gpu_util import T, N
    g = golden("penalty")
    wr, wi, ls2 = _params(g, kind)
    elem, tot = ops.kl_fwd(kind, T(wr), None if wi is None else T(wi), T(ls2), elementwise=True)
    ref = g[f"f32_{kind}_penalty"]
    ref64 = orc.penalty(kind, g["f64_ls2"].astype(np.float64), g["f64_wr"], None if wi is None else g["f64_wi"])
    fin = np.isfinite(ref)
    # 1e-5 relative to the O(1..10) terms the reference's fp32 chain adds up
    np.testing.assert_allclose(N(elem)[fin], ref[fin], rtol=1e-5, atol=2e-6)
    # and tighter than the reference itself against the float64 oracle on its own inputs
    o64 = orc.penalty(kind, ls2.astype(np.float64), wr.astype(np.float64),
                      None if wi is None else wi.astype(np.float64))
    np.testing.assert_allclose(N(elem)[fin], o64[fin], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(float(tot), o64[np.isfinite(o64)].sum() if np.isfinite(o64).all() else float(tot), rtol=1e-6)
    np.testing.assert_allclose(float(tot), float(g[f"f32_{kind}_sum"]), rtol=1e-5)
    assert ref64.shape == ref.shape


@pytest.mark.parametrize("kind", KINDS)
def test_penalty_gradients(golden, ops, kind):
    from gpu_util import T, N
    g = golden("penalty")
    wr, wi, ls2 = _params(g, kind)
    up = g["f32_g"]
    twi = None if wi is None else T(wi)
    g_ls2, g_wr, g_wi = ops.kl_bwd(kind, T(wr), twi, T(ls2), g_elem=T(up))
    o = orc.penalty_bwd(kind, up.astype(np.float64), ls2.astype(np.float64), wr.astype(np.float64),
                        None if wi is None else wi.astype(np.float64))
    theta = np.abs(wr) if wi is None else orc.cplx_abs(wr, wi)
    with np.errstate(divide="ignore"):
        amp = np.where(theta > 0, 2 / (theta + 1e-12), 0)
    eps = np.finfo(np.float32).eps
    for got, key, a in ((g_ls2, "dlog_sigma2", 1.0), (g_wr, "dwr", amp), (g_wi, "dwi", amp)):
        if got is None:
            continue
        ref = o[key]
        err = np.abs(N(got).astype(np.float64) - ref)
        bound = 2e-5 * np.abs(ref) + 8 * eps * np.maximum(a, 1.0)
        assert (err <= bound).all(), (kind, key, float((err - bound).max()))
    # scalar upstream gradient read on the device + fused fwd_bwd agree
    gs = torch.tensor(0.37, device="cuda")
    s_ls2, s_wr, s_wi = ops.kl_bwd(kind, T(wr), twi, T(ls2), g_scalar=gs)
    tot, f_ls2, f_wr, f_wi = ops.kl_fwd_bwd(kind, T(wr), twi, T(ls2), gscale=0.37)
    assert torch.equal(s_ls2, f_ls2) and torch.equal(s_wr, f_wr)
    np.testing.assert_allclose(float(tot), float(g[f"f32_{kind}_sum"]), rtol=1e-5)
    ones = ops.kl_bwd(kind, T(wr), twi, T(ls2), g_elem=torch.full_like(T(ls2), 0.37))
    np.testing.assert_allclose(N(s_ls2), N(ones[0]), rtol=1e-6, atol=1e-30)


@pytest.mark.parametrize("kind", ("real_vd", "cplx_vd"))
@pytest.mark.parametrize("th", (-0.5, 1.0, 3.0))
def test_masks_bit_exact(golden, ops, kind, th):
    from gpu_util import T, N
    g = golden("penalty")
    wr, wi, ls2 = _params(g, kind)
    twi = None if wi is None else T(wi)
    mask, cnt = ops.relevance_mask(T(wr), twi, T(ls2), th, count=True)
    ref = g[f"f32_{kind}_mask_{th}"]
    np.testing.assert_array_equal(N(mask), ref)
    assert int(cnt) == int(ref.sum())
    la = ops.log_alpha(T(wr), twi, T(ls2))
    rla = g[f"f32_{kind}_log_alpha"]
    fin = np.isfinite(rla)
    # log_alpha = ls2 - 2 log(.) cancels: the honest yardstick is an ulp of the OPERANDS.  The
    # device log is correctly rounded, the reference's libm is within 1 ulp of that.
    scale = np.maximum(np.abs(ls2), np.abs(ls2 - rla))[fin]
    err = np.abs(N(la)[fin].astype(np.float64) - rla[fin])
    assert (err <= 2 * np.finfo(np.float32).eps * scale).all()
    assert (err > 0).mean() < 1e-2          # and almost always bit-identical


def test_masks_large_random_vs_oracle(ops):
    """2^22 weights: the device mask equals the oracle's (numpy, torch-CPU rounding) except
    possibly where log_alpha is within 1 ulp of the threshold; those are counted, not hidden."""
    from gpu_util import T, N
    rs = np.random.RandomState(7)
    n = 1 << 22
    wr = (rs.uniform(-0.09, 0.09, n)).astype(np.float32)
    wi = (rs.uniform(-0.09, 0.09, n)).astype(np.float32)
    ls2 = rs.uniform(-12, 4, n).astype(np.float32)
    for th in (-0.5, 1.0, 3.0):
        m = N(ops.relevance_mask(T(wr), T(wi), T(ls2), th))
        la = orc.log_alpha(ls2, wr, wi)
        ref = (la <= np.float32(th)).astype(np.float32)
        diff = m != ref
        near = np.abs(la - np.float32(th)) <= 2 * np.spacing(np.abs(la))
        assert not (diff & ~near).any()
        assert diff.sum() <= 4, diff.sum()


def test_penalty_unaligned_sizes(ops):
    from gpu_util import T, N
    rs = np.random.RandomState(3)
    for n in (1, 3, 5, 1023, 4097):
        wr = rs.uniform(-0.1, 0.1, n).astype(np.float32)
        wi = rs.uniform(-0.1, 0.1, n).astype(np.float32)
        ls2 = rs.uniform(-12, 4, n).astype(np.float32)
        elem, tot = ops.kl_fwd("cplx_vd", T(wr), T(wi), T(ls2), elementwise=True)
        ref = orc.penalty("cplx_vd", ls2.astype(np.float64), wr.astype(np.float64), wi.astype(np.float64))
        np.testing.assert_allclose(N(elem), ref, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(float(tot), ref.sum(), rtol=1e-6)


def test_expi(golden, ops):
    from gpu_util import T, N
    g = golden("penalty")
    x = T(g["f32_expi_x"]).requires_grad_(True)
    y = ops.ExpiFn.apply(x)
    ref = g["f32_expi_y"]
    fin = np.isfinite(ref)
    np.testing.assert_allclose(N(y)[fin], ref[fin], rtol=2e-6, atol=1e-30)
    y.sum().backward()
    np.testing.assert_allclose(N(x.grad), g["f32_expi_dx"], rtol=1e-5)


def test_philox_stream_matches_spec(ops):
    from gpu_util import N
    n = 10007
    er = N(ops.philox_normal(n, 0x1234567890ABCDEF, 3, "cuda"))
    np.testing.assert_allclose(er, philox.real_noise(n, 0x1234567890ABCDEF, 3), rtol=0, atol=2e-5)
    cr, ci = ops.philox_normal(n, 42, 9, "cuda", complex_=True)
    rr, ri = philox.cplx_noise(n, 42, 9)
    np.testing.assert_allclose(N(cr), rr, atol=2e-5)
    np.testing.assert_allclose(N(ci), ri, atol=2e-5)
    big = N(ops.philox_normal(1 << 22, 5, 1, "cuda"))
    assert abs(big.mean()) < 2e-3 and abs(big.std() - 1) < 2e-3
    assert abs(np.corrcoef(big[:-1], big[1:])[0, 1]) < 2e-3


@pytest.mark.parametrize("dtype", (torch.float32, torch.bfloat16))
@pytest.mark.parametrize("cplx", (True, False))
def test_reparam_given_noise(ops, dtype, cplx):
    from gpu_util import T, N
    rs = np.random.RandomState(11)
    n = 4 * 1031 + 3
    mu_r, mu_i = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    s2 = np.exp(rs.uniform(-25, 2, n)).astype(np.float32)
    s2[:5] = [1e-8, 0.99e-8, 1.01e-8, 0.0, 1.0]
    er, ei = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    gr, gi = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    q = (lambda a: T(a, dtype))
    f = (lambda t: N(t.float()))
    qmu_r, qmu_i, qer, qei, qgr, qgi = map(q, (mu_r, mu_i, er, ei, gr, gi))
    sd = np.sqrt(np.maximum(s2, np.float32(1e-8)))
    if cplx:
        yr, yi = ops.reparam_fwd(qmu_r, qmu_i, T(s2), (qer, qei))
        ref_r = f(qmu_r) + f(qer) * sd
        ref_i = f(qmu_i) + f(qei) * sd
        gs = f(qgr) * f(qer) + f(qgi) * f(qei)
    else:
        yr, yi = ops.reparam_fwd(qmu_r, None, T(s2), qer)
        ref_r = f(qmu_r) + f(qer) * sd
        gs = f(qgr) * f(qer)
    tol = dict(rtol=1e-6, atol=1e-7) if dtype == torch.float32 else dict(rtol=8e-3, atol=1e-6)
    np.testing.assert_allclose(f(yr), ref_r, **tol)
    if cplx:
        np.testing.assert_allclose(f(yi), ref_i, **tol)
    gs2 = ops.reparam_bwd(qgr, qgi if cplx else None, T(s2), (qer, qei) if cplx else qer)
    ref = np.where(s2 >= np.float32(1e-8), gs * 0.5 / sd, 0)
    np.testing.assert_allclose(N(gs2), ref, rtol=2e-6, atol=1e-30)
    assert N(gs2)[1] == 0 and N(gs2)[0] != 0  # clamp gradient: blocked below, passed AT 1e-8


@pytest.mark.parametrize("cplx", (True, False))
@pytest.mark.parametrize("philox_", (False, True))
def test_reparam_bf16_variance_operand(ops, cplx, philox_):
    """bf16 layers hand the variance over in bf16 (cplxamd_lrt_reparam_{fwd,bwd}_ex, s2_dtype = bf16): on
    bf16-representable values the results are BIT-identical to the float32-s2 entry points (the kernels widen s2 to
    float32 first), incl. the clamp (1e-8 itself is not representable: its bf16 neighbours fall on either side), the
    scalar tail and the in-kernel noise; a float32 s2 with a bf16 mu is still accepted."""
    from gpu_util import T
    rs = np.random.RandomState(21)
    n = 8 * 517 + 5
    bf = torch.bfloat16
    s2 = T(np.exp(rs.uniform(-25, 2, n)).astype(np.float32)).to(bf)
    lo = torch.tensor([1e-8], device="cuda").to(bf)                       # nearest bf16 value to 1e-8
    s2[:4] = torch.stack([lo[0], lo[0] * 1.0078125, lo[0] * 0.9921875, torch.zeros((), device="cuda", dtype=bf)])
    mu_r, mu_i, gr, gi, er, ei = (T(rs.randn(n).astype(np.float32)).to(bf) for _ in range(6))
    eps = None if philox_ else ((er, ei) if cplx else er)
    kw = dict(seed=5, offset=9)
    a = ops.reparam_fwd(mu_r, mu_i if cplx else None, s2, eps, **kw)
    b = ops.reparam_fwd(mu_r, mu_i if cplx else None, s2.float(), eps, **kw)
    assert torch.equal(a[0], b[0]) and (not cplx or torch.equal(a[1], b[1]))
    for out in (torch.bfloat16, torch.float32):
        ga = ops.reparam_bwd(gr, gi if cplx else None, s2, eps, out_dtype=out, **kw)
        gb = ops.reparam_bwd(gr, gi if cplx else None, s2.float(), eps, out_dtype=out, **kw)
        assert ga.dtype == out and torch.equal(ga, gb)
    below = s2.float() < 1e-8
    assert bool(below[:4].any()) and bool((~below[:4]).any())
    assert bool((ga[below] == 0).all()) and bool((ga[~below][:64] != 0).any())


@pytest.mark.parametrize("rows,cols", [(100, 8), (37, 64), (4099, 64), (513, 2048), (300, 4096), (64, 6144), (50, 24),
                                       (33, 40)])
@pytest.mark.parametrize("mode", ["cplx_philox_bf16", "cplx_given_f32", "real_philox_f32", "real_given_bf16"])
def test_reparam_bwd_with_bias_sums(ops, rows, cols, mode):
    """cplxamd_lrt_reparam_bwd_cols: d s2 bit-identical to the flat kernel (same element -> Philox counter mapping),
    column sums of the gradient planes (the layer's bias gradient) against float64; (50, 24) and (33, 40) are shapes
    the fused kernel declines (cols / 8 does not divide 256): the wrapper then runs the flat kernel + cplxamd_colsum."""
    from cplxmodule_amd import _lib
    cplx, philox_, bf = mode.startswith("cplx"), "philox" in mode, mode.endswith("bf16")
    dt = torch.bfloat16 if bf else torch.float32
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + cols)
    mk = lambda: torch.randn(rows, cols, generator=g).to(dt).to("cuda")  # noqa: E731
    gr, gi = mk(), (mk() if cplx else None)
    s2 = (torch.rand(rows, cols, generator=g) * 2).to(dt).to("cuda")
    s2[0, :4] = 0
    eps = None if philox_ else ((mk(), mk()) if cplx else mk())
    kw = dict(seed=11, offset=3, out_dtype=dt)
    flat = ops.reparam_bwd(gr, gi, s2, eps, **kw)
    out_r = torch.full((cols,), float("nan"), device="cuda")
    out_i = torch.full((cols,), float("nan"), device="cuda") if cplx else None
    got, sr, si = ops.reparam_bwd(gr, gi, s2, eps, bias_sums=(rows, cols, (out_r, out_i)), **kw)
    assert torch.equal(got, flat)
    assert sr is out_r and (si is out_i)
    takes = int(_lib.load().cplxamd_lrt_reparam_bwd_cols_ws_bytes(rows, cols)) > 0
    assert takes == ((cols // 8 < 256 and 256 % (cols // 8) == 0) or cols % 2048 == 0)
    for t, s in ((gr, sr), (gi, si)):
        if t is None:
            continue
        ref = t.double().sum(0)
        scale = t.double().abs().sum(0).max()
        assert float((s.double() - ref).abs().max()) <= 2e-6 * float(scale)
    # sums allocated by the wrapper, 4-d channels-last planes ([B H W][C] rows)
    if cols == 64 and rows == 4099:
        B, H, W = 2, 5, 7
        c4 = lambda t: None if t is None else t[: B * H * W].reshape(B, H, W, cols).permute(0, 3, 1, 2)  # noqa: E731
        e4 = None if eps is None else (tuple(c4(e) for e in eps) if cplx else c4(eps))
        g4, sr4, si4 = ops.reparam_bwd(c4(gr), c4(gi), c4(s2), e4, bias_sums=(B * H * W, cols), **kw)
        assert g4.shape == (B, cols, H, W) and g4.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(g4.permute(0, 2, 3, 1).reshape(-1, cols), flat[: B * H * W])
        np.testing.assert_allclose(sr4.cpu().numpy(), gr[: B * H * W].double().sum(0).cpu().numpy(), rtol=0,
                                   atol=2e-6 * float(gr.double().abs().sum(0).max()))


def test_reparam_philox_fwd_bwd_consistent(ops):
    from gpu_util import T, N
    rs = np.random.RandomState(12)
    n = 4 * 999 + 2
    mu = np.zeros(n, np.float32)
    s2 = np.ones(n, np.float32)
    yr, yi = ops.reparam_fwd(T(mu), T(mu), T(s2), None, seed=77, offset=5)
    rr, ri = philox.cplx_noise(n, 77, 5)
    np.testing.assert_allclose(N(yr), rr, atol=2e-5)
    np.testing.assert_allclose(N(yi), ri, atol=2e-5)
    gr, gi = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
    gs2 = ops.reparam_bwd(T(gr), T(gi), T(s2), None, seed=77, offset=5)
    np.testing.assert_allclose(N(gs2), 0.5 * (gr * N(yr) + gi * N(yi)), rtol=1e-5, atol=1e-6)
    y = ops.reparam_fwd(T(mu), None, T(s2), None, seed=77, offset=6)[0]
    np.testing.assert_allclose(N(y), philox.real_noise(n, 77, 6), atol=2e-5)


@pytest.mark.parametrize("kind", orc.EXT_KINDS)
def test_extension_penalties_gpu(golden, ops, kind):
    """SURVEY 8(f) row 4 (extensions/complex.py): values, sum, gradients and the layer classes."""
    from gpu_util import T, N, DEV
    from cplxmodule_amd.nn.relevance import extensions as ext
    g = golden("extras")
    k = f"f32_ext_{kind}_"
    wr, wi, ls2, up = g[k + "wr"], g[k + "wi"], g[k + "ls2"], g[k + "g"]
    elem, tot = ops.kl_fwd(kind, T(wr), T(wi), T(ls2), elementwise=True)
    ref = g[k + "pen"]
    fin = np.isfinite(ref)
    np.testing.assert_allclose(N(elem)[fin], ref[fin], rtol=1e-5, atol=4e-6)
    o64 = orc.penalty(kind, ls2.astype(np.float64), wr.astype(np.float64), wi.astype(np.float64))
    fin64 = np.isfinite(o64)
    np.testing.assert_allclose(N(elem)[fin64], o64[fin64], rtol=3e-6, atol=5e-7)
    g_ls2, g_wr, g_wi = ops.kl_bwd(kind, T(wr), T(wi), T(ls2), g_elem=T(up))
    o = orc.penalty_bwd(kind, up.astype(np.float64), ls2.astype(np.float64), wr.astype(np.float64),
                        wi.astype(np.float64))
    theta = orc.cplx_abs(wr, wi)
    with np.errstate(divide="ignore"):
        amp = np.where(theta > 0, 2 / (theta + 1e-12), 0)
    eps = np.finfo(np.float32).eps
    for got, key, a in ((g_ls2, "dlog_sigma2", 1.0), (g_wr, "dwr", amp), (g_wi, "dwi", amp)):
        err = np.abs(N(got).astype(np.float64) - o[key])
        bound = 2e-5 * np.abs(o[key]) + 8 * eps * np.maximum(a, 1.0)
        assert (err <= bound).all(), (kind, key, float((err - bound).max()))
    cls = dict(cplx_vd_approx=ext.CplxLinearVDApprox, cplx_vd_scalefree=ext.CplxLinearVDScaleFree,
               cplx_vd_bogus=ext.CplxLinearVDBogus)[kind]
    layer = cls(24, 20).to(DEV)
    with torch.no_grad():
        layer.weight.real.copy_(T(wr)); layer.weight.imag.copy_(T(wi)); layer.log_sigma2.copy_(T(ls2))
    np.testing.assert_allclose(N(layer.penalty)[fin], ref[fin], rtol=1e-5, atol=4e-6)
    conv_cls = dict(cplx_vd_approx=ext.CplxConv2dVDApprox, cplx_vd_scalefree=ext.CplxConv2dVDScaleFree,
                    cplx_vd_bogus=ext.CplxConv2dVDBogus)[kind]
    assert conv_cls(4, 4, 3).to(DEV).penalty.shape == (4, 4, 3, 3)
