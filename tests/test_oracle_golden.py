"""The numpy oracle (oracle/cplx_oracle.py) against outputs of the REAL
reference stored in tests/golden/ (made by oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import TOL
from oracle import cplx_oracle as orc
from oracle.gen_golden_cases import CONV_CASES

TAGS = ("f32", "f64")


def close(a, b, tag, scale=1.0):
    t = TOL[tag]
    np.testing.assert_allclose(a, b, rtol=t["rtol"] * scale, atol=t["atol"] * scale)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", "abc")
def test_linear(golden, tag, case):
    g = golden("linear")
    k = f"{tag}_{case}_"
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi", "br", "bi")]
    for algo, ref in (("4m", "naive"), ("3m", "3m"), ("cat", "cat")):
        yr, yi = orc.cplx_linear(*a, algo=algo)
        close(yr, g[k + f"y_{ref}_r"], tag, 10)
        close(yi, g[k + f"y_{ref}_i"], tag, 10)
    yr, yi = orc.cplx_linear(*a[:4])
    close(yr, g[k + "y_nobias_r"], tag, 10)
    bw = orc.cplx_linear_bwd(g[k + "gr"], g[k + "gi"], *a[:4])
    for n in ("dxr", "dxi", "dwr", "dwi", "dbr", "dbi"):
        close(bw[n], g[k + n], tag, 30)


@pytest.mark.parametrize("tag", TAGS)
def test_matmul(golden, tag):
    g = golden("linear")
    k = f"{tag}_mm_"
    mr, mi = orc.cplx_matmul(g[k + "ur"], g[k + "ui"], g[k + "vr"], g[k + "vi"])
    close(mr, g[k + "mr"], tag, 10)
    close(mi, g[k + "mi"], tag, 10)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_cplx_linear(golden, tag):
    g = golden("lrt_linear")
    k = f"{tag}_cplx_"
    er, ei = orc.cplx_randn_from_tape(g[k + "tape"])
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi", "br", "bi", "ls2")]
    yr, yi, _ = orc.lrt_cplx_linear(*a, er, ei)
    close(yr, g[k + "yr"], tag, 10)
    close(yi, g[k + "yi"], tag, 10)
    mur, mui = orc.cplx_linear(*a[:6])
    close(mur, g[k + "yr_eval"], tag, 10)
    bw = orc.lrt_cplx_linear_bwd(g[k + "gr"], g[k + "gi"], *a[:4], a[6], er, ei)
    for n, m in (("dxr", "dxr"), ("dxi", "dxi"), ("dwr", "dwr"), ("dwi", "dwi"),
                 ("dbr", "dbr"), ("dbi", "dbi"), ("dlog_sigma2", "dls2")):
        close(bw[n], g[k + m], tag, 100)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_clamp_boundary(golden, tag):
    g = golden("lrt_linear")
    k = f"{tag}_cplx_"
    er, ei = orc.cplx_randn_from_tape(g[k + "clamp_tape"])
    a = [g[k + n] for n in ("clamp_xr", "clamp_xi", "wr", "wi", "br", "bi", "clamp_ls2")]
    yr, yi, aux = orc.lrt_cplx_linear(*a, er, ei)
    assert (aux["s2"] < 1e-8).any() and (aux["s2"] > 1e-8).any()
    close(yr, g[k + "clamp_yr"], tag, 10)
    bw = orc.lrt_cplx_linear_bwd(g[k + "gr"], g[k + "gi"], *a[:4], a[6], er, ei)
    close(bw["dlog_sigma2"], g[k + "clamp_dls2"], tag, 100)
    close(bw["dxr"], g[k + "clamp_dxr"], tag, 100)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_real_linear(golden, tag):
    g = golden("lrt_linear")
    k = f"{tag}_real_"
    y, _ = orc.lrt_real_linear(g[k + "x"], g[k + "w"], g[k + "b"], g[k + "ls2"], g[k + "eps"])
    close(y, g[k + "y"], tag, 10)
    bw = orc.lrt_real_linear_bwd(g[k + "g"], g[k + "x"], g[k + "w"], g[k + "ls2"], g[k + "eps"])
    for n, m in (("dx", "dx"), ("dw", "dw"), ("db", "db"), ("dlog_sigma2", "dls2")):
        close(bw[n], g[k + m], tag, 100)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", orc.KINDS)
def test_penalty(golden, tag, kind):
    g = golden("penalty")
    k = f"{tag}_"
    wr, ls2 = g[k + "wr"], g[k + "ls2"]
    wi = g[k + "wi"] if kind.startswith("cplx") else None
    la = orc.log_alpha(ls2, wr, wi)
    fin = np.isfinite(g[k + kind + "_log_alpha"])
    np.testing.assert_array_equal(np.isfinite(la), fin)
    close(la[fin], g[k + kind + "_log_alpha"][fin], tag)
    pen = orc.penalty(kind, ls2, wr, wi)
    ref = g[k + kind + "_penalty"]
    fin = np.isfinite(ref)
    close(pen[fin], ref[fin], tag, 4)
    close(pen[fin].sum(), ref[fin].sum(), tag, 4)
    bw = orc.penalty_bwd(kind, g[k + "g"], ls2, wr, wi)
    # The reference forms f'(t) by autograd as a sum of O(1) terms (e.g.
    # 1 - exp(-e^t)), so its f' carries an ABSOLUTE error of a few ulp(1); the
    # weight gradient multiplies that by amp = 2|w| / (theta (theta + 1e-12)),
    # which is huge for tiny weights.  The oracle evaluates f' without the
    # cancellation, hence the amplification-aware tolerance.
    eps = np.finfo(ls2.dtype).eps
    theta = np.abs(wr) if wi is None else orc.cplx_abs(wr, wi)
    with np.errstate(divide="ignore", invalid="ignore"):
        amp = np.where(theta > 0, 2 / (theta + 1e-12), 0)
    rt = 2e-5 if tag == "f32" else 1e-10
    for n, m, a in (("dlog_sigma2", "dls2", 1.0), ("dwr", "dwr", amp), ("dwi", "dwi", amp)):
        if n in bw:
            r = g[k + kind + "_" + m]
            ok = np.isfinite(r)
            err = np.abs(bw[n] - r)
            bound = rt * np.abs(r) + 8 * eps * np.maximum(a, 1.0)
            assert (err[ok] <= bound[ok] if np.ndim(bound) else err[ok] <= bound).all(), (n, err[ok].max())


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", orc.KINDS)
@pytest.mark.parametrize("th", (-0.5, 1.0, 3.0))
def test_masks_bit_exact(golden, tag, kind, th):
    g = golden("penalty")
    k = f"{tag}_"
    wi = g[k + "wi"] if kind.startswith("cplx") else None
    m = orc.relevance_mask(th, g[k + "ls2"], g[k + "wr"], wi)
    assert int(g[k + kind + f"_near_{th}"]) == 0
    np.testing.assert_array_equal(m, g[k + kind + f"_mask_{th}"])
    assert 0 < m.sum() < m.size


@pytest.mark.parametrize("tag", TAGS)
def test_expi(golden, tag):
    g = golden("penalty")
    x = g[f"{tag}_expi_x"]
    close(orc.expi(x), g[f"{tag}_expi_y"], tag)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", list(CONV_CASES))
def test_conv2d(golden, tag, case):
    g = golden("conv")
    B, Ci, Co, H, W, ks, st, pd, dl, gp, mode = CONV_CASES[case]
    k = f"{tag}_{case}_"
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi", "br", "bi")]
    yr, yi = orc.cplx_conv2d(*a, stride=st, padding=pd, dilation=dl, groups=gp, padding_mode=mode)
    close(yr, g[k + "yr"], tag, 10)
    close(yi, g[k + "yi"], tag, 10)
    if mode == "zeros":
        bw = orc.cplx_conv2d_bwd(g[k + "gr"], g[k + "gi"], *a[:4], stride=st, padding=pd,
                                 dilation=dl, groups=gp)
        for n in ("dxr", "dxi", "dwr", "dwi", "dbr", "dbi"):
            close(bw[n], g[k + n], tag, 50)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_conv(golden, tag):
    g = golden("conv")
    k = f"{tag}_lrtc_"
    er, ei = orc.cplx_randn_from_tape(g[k + "tape"])
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi", "br", "bi", "ls2")]
    yr, yi, _ = orc.lrt_cplx_conv2d(*a, er, ei, stride=1, padding=1)
    close(yr, g[k + "yr"], tag, 10)
    close(yi, g[k + "yi"], tag, 10)
    bw = orc.lrt_cplx_conv2d_bwd(g[k + "gr"], g[k + "gi"], *a[:4], a[6], er, ei, stride=1, padding=1)
    for n, m in (("dxr", "dxr"), ("dxi", "dxi"), ("dwr", "dwr"), ("dwi", "dwi"),
                 ("dbr", "dbr"), ("dlog_sigma2", "dls2")):
        close(bw[n], g[k + m], tag, 100)
    close(orc.penalty("cplx_vd", a[6], a[2], a[3]).sum(), g[k + "penalty_sum"], tag, 10)
    k = f"{tag}_lrtr_"
    y, _ = orc.lrt_real_conv2d(g[k + "x"], g[k + "w"], g[k + "b"], g[k + "ls2"], g[k + "eps"],
                               stride=2, padding=1)
    close(y, g[k + "y"], tag, 10)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ("2d", "1d", "1d3"))
def test_batchnorm(golden, tag, name):
    g = golden("batchnorm")
    k = f"{tag}_{name}_"
    W, b = g[k + "weight"], g[k + "bias"]
    F_ = W.shape[-1]
    rm = np.zeros((2, F_), W.dtype)
    rv = np.stack([np.ones(F_), np.zeros(F_), np.zeros(F_), np.ones(F_)]).reshape(2, 2, F_).astype(W.dtype)
    sc = 200 if tag == "f32" else 1e4
    for step in range(3):
        s = k + f"s{step}_"
        yr, yi = orc.cplx_batch_norm(g[s + "xr"], g[s + "xi"], rm, rv, W, b, True, 0.1, 1e-5)
        close(yr, g[s + "yr"], tag, sc)
        close(yi, g[s + "yi"], tag, sc)
        close(rm, g[s + "running_mean"], tag, 10)
        close(rv, g[s + "running_var"], tag, 10)
        bw = orc.cplx_batch_norm_bwd(g[s + "gr"], g[s + "gi"], g[s + "xr"], g[s + "xi"], None, None,
                                     W, True, 1e-5)
        for n in ("dxr", "dxi", "dweight", "dbias"):
            close(bw[n], g[s + n], tag, sc)
    yr, yi = orc.cplx_batch_norm(g[s + "xr"], g[s + "xi"], rm, rv, W, b, False, 0.1, 1e-5)
    close(yr, g[k + "eval_yr"], tag, sc)
    bw = orc.cplx_batch_norm_bwd(g[s + "gr"], g[s + "gi"], g[s + "xr"], g[s + "xi"], rm, rv, W,
                                 False, 1e-5)
    for n in ("dxr", "dxi", "dweight", "dbias"):
        close(bw[n], g[k + "eval_" + n], tag, sc)


@pytest.mark.parametrize("tag", TAGS)
def test_batchnorm_functional(golden, tag):
    g = golden("batchnorm")
    k = f"{tag}_func_"
    yr, yi = orc.cplx_batch_norm(g[k + "xr"], g[k + "xi"], None, None, None, None, True, 0.1, 1e-3)
    close(yr, g[k + "yr"], tag, 100)
    bw = orc.cplx_batch_norm_bwd(g[k + "gr"], g[k + "gi"], g[k + "xr"], g[k + "xi"], None, None,
                                 None, True, 1e-3)
    close(bw["dxr"], g[k + "dxr"], tag, 100)
    close(bw["dxi"], g[k + "dxi"], tag, 100)


# ---- converters, modReLU, dropout ------------------------------------- #
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_oracle_layout_converters(golden, tag):
    g = golden("extras")
    x = g[f"{tag}_il_x"]
    re, im = orc.from_interleaved_real(x)
    assert np.array_equal(re, g[f"{tag}_il_re"]) and np.array_equal(im, g[f"{tag}_il_im"])
    assert np.array_equal(orc.to_interleaved_real(re, im), g[f"{tag}_il_back"])
    assert np.array_equal(orc.to_interleaved_real(re, im, flatten=False), g[f"{tag}_il_stack"])
    cr, ci = orc.from_concatenated_real(x)
    assert np.array_equal(cr, g[f"{tag}_cat_re"]) and np.array_equal(ci, g[f"{tag}_cat_im"])
    assert np.array_equal(np.concatenate([cr, ci], -1), g[f"{tag}_cat_back"])


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("case", ["scalar", "one", "chan"])
def test_oracle_modrelu(golden, tag, case):
    g = golden("extras")
    zr, zi, gr, gi = (g[f"{tag}_mr_{k}"] for k in ("zr", "zi", "gr", "gi"))
    k = f"{tag}_mr_{case}_"
    tau = g[k + "tau"].astype(zr.dtype)
    tau_b = tau if tau.ndim != 1 or tau.shape[0] != 1 else tau[0]
    yr, yi = orc.modrelu(zr, zi, tau_b)
    tol = dict(rtol=1e-6, atol=1e-7) if tag == "f32" else dict(rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(yr, g[k + "yr"], **tol)
    np.testing.assert_allclose(yi, g[k + "yi"], **tol)
    bw = orc.modrelu_bwd(gr, gi, zr, zi, tau_b)
    gtol = dict(rtol=2e-5, atol=2e-5) if tag == "f32" else dict(rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(bw["dzr"], g[k + "dzr"], **gtol)
    np.testing.assert_allclose(bw["dzi"], g[k + "dzi"], **gtol)
    if case != "scalar":
        want = g[k + "dtau"]
        got = bw["dtau"].sum() if want.size == 1 else bw["dtau"].sum(axis=(0, 2), keepdims=False).reshape(want.shape)
        np.testing.assert_allclose(np.asarray(got).reshape(want.shape), want,
                                   **(dict(rtol=1e-4, atol=1e-4) if tag == "f32" else gtol))


def test_reference_dropout_contract(golden):
    """What the reference's CplxDropout guarantees (the property our Philox version must share):
    real and imaginary parts share their fate, survivors are scaled by 1 / (1 - p)."""
    g = golden("extras")
    zr, zi, yr, yi = g["do_zr"], g["do_zi"], g["do_yr"], g["do_yi"]
    kept_r, kept_i = yr != 0, yi != 0
    assert np.array_equal(kept_r, kept_i)
    np.testing.assert_allclose(yr[kept_r], zr[kept_r] / 0.7, rtol=1e-6)
    np.testing.assert_allclose(yi[kept_i], zi[kept_i] / 0.7, rtol=1e-6)
    assert 0.6 < kept_r.mean() < 0.8


def test_oracle_dropout_mask_rate():
    m = orc.cplx_dropout_mask(200003, 0.3, seed=7, offset=3)
    assert m.shape == (200003,) and abs(m.mean() - 0.7) < 5e-3
    assert not np.array_equal(m, orc.cplx_dropout_mask(200003, 0.3, seed=7, offset=4))


POOLS = {"k2": dict(kernel_size=2), "k3s2p1": dict(kernel_size=3, stride=2, padding=1),
         "rect": dict(kernel_size=(3, 2), stride=(2, 1), padding=(1, 0), dilation=(1, 2), ceil_mode=True)}


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", list(POOLS))
def test_oracle_max_pool2d(golden, tag, name):
    g = golden("extras")
    zr, zi = g[f"{tag}_mp_zr"], g[f"{tag}_mp_zi"]
    k = f"{tag}_mp_{name}_"
    yr, yi, idx = orc.cplx_max_pool2d(zr, zi, **POOLS[name])
    assert np.array_equal(yr, g[k + "yr"]) and np.array_equal(yi, g[k + "yi"])
    dzr, dzi = orc.cplx_max_pool2d_bwd(g[k + "gr"], g[k + "gi"], idx, zr.shape)
    tol = dict(rtol=1e-6, atol=1e-6) if tag == "f32" else dict(rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dzr, g[k + "dzr"], **tol)
    np.testing.assert_allclose(dzi, g[k + "dzi"], **tol)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", orc.EXT_KINDS)
def test_extension_penalties(golden, tag, kind):
    """CplxLinearVDApprox / VDScaleFree penalties of the reference."""
    g = golden("extras")
    k = f"{tag}_ext_{kind}_"
    wr, wi, ls2, up = g[k + "wr"], g[k + "wi"], g[k + "ls2"], g[k + "g"]
    pen, ref = orc.penalty(kind, ls2, wr, wi), g[k + "pen"]
    fin = np.isfinite(ref)
    np.testing.assert_array_equal(np.isfinite(pen), fin)
    close(pen[fin], ref[fin], tag, 8)
    bw = orc.penalty_bwd(kind, up, ls2, wr, wi)
    eps = np.finfo(ls2.dtype).eps
    theta = orc.cplx_abs(wr, wi)
    with np.errstate(divide="ignore", invalid="ignore"):
        amp = np.where(theta > 0, 2 / (theta + 1e-12), 0)
    rt = 2e-5 if tag == "f32" else 1e-10
    for n, m, a in (("dlog_sigma2", "dls2", 1.0), ("dwr", "dwr", amp), ("dwi", "dwi", amp)):
        r = g[k + m]
        ok = np.isfinite(r)
        err = np.abs(bw[n] - r)
        bound = rt * np.abs(r) + 8 * eps * np.maximum(a, 1.0)
        assert (err[ok] <= (bound[ok] if np.ndim(bound) else bound)).all(), (n, err[ok].max())


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("mode", ["conj", "plain"])
def test_bilinear(golden, tag, mode):
    g = golden("bilinear")
    k = f"{tag}_fn_{mode}_"
    a = [g[k + n] for n in ("x1r", "x1i", "x2r", "x2i", "wr", "wi")]
    yr, yi = orc.cplx_bilinear(*a, g[k + "br"], g[k + "bi"], conjugate=mode == "conj")
    close(yr, g[k + "yr"], tag, 30)
    close(yi, g[k + "yi"], tag, 30)
    bw = orc.cplx_bilinear_bwd(g[k + "gr"], g[k + "gi"], *a, conjugate=mode == "conj")
    for n in ("dx1r", "dx1i", "dx2r", "dx2i", "dwr", "dwi", "dbr", "dbi"):
        close(bw[n], g[k + n], tag, 100)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_bilinear(golden, tag):
    g = golden("bilinear")
    k = f"{tag}_vd_"
    er, ei = orc.cplx_randn_from_tape(g[k + "tape"])
    a = [g[k + n] for n in ("x1r", "x1i", "x2r", "x2i", "wr", "wi")]
    yr, yi, _ = orc.lrt_cplx_bilinear(*a, g[k + "br"], g[k + "bi"], g[k + "ls2"], er, ei)
    close(yr, g[k + "yr"], tag, 30)
    close(yi, g[k + "yi"], tag, 30)
    mur, mui = orc.cplx_bilinear(*a, g[k + "br"], g[k + "bi"])
    close(mur, g[k + "yr_eval"], tag, 30)
    close(mui, g[k + "yi_eval"], tag, 30)
    bw = orc.lrt_cplx_bilinear_bwd(g[k + "gr"], g[k + "gi"], *a, g[k + "ls2"], er, ei)
    for n in ("dx1r", "dx1i", "dx2r", "dx2i", "dwr", "dwi", "dbr", "dbi"):
        close(bw[n], g[k + n], tag, 300)
    close(bw["dlog_sigma2"], g[k + "dls2"], tag, 300)
    close(orc.penalty("cplx_vd", g[k + "ls2"], g[k + "wr"], g[k + "wi"]), g[k + "pen"], tag, 30)
    # real layer
    k = f"{tag}_real_"
    y, _ = orc.lrt_real_bilinear(g[k + "x1"], g[k + "x2"], g[k + "w"], g[k + "b"], g[k + "ls2"], g[k + "eps"])
    close(y, g[k + "y"], tag, 30)
    close(orc.real_bilinear(g[k + "x1"], g[k + "x2"], g[k + "w"], g[k + "b"]), g[k + "y_eval"], tag, 30)
    bw = orc.lrt_real_bilinear_bwd(g[k + "g"], g[k + "x1"], g[k + "x2"], g[k + "w"], g[k + "ls2"], g[k + "eps"])
    for n, m in (("dx1", "dx1"), ("dx2", "dx2"), ("dw", "dw"), ("db", "db"), ("dlog_sigma2", "dls2")):
        close(bw[n], g[k + m], tag, 300)


from oracle.gen_golden_cases import CONV3D_CASES, POOL3D_CASES  # noqa: E402


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", list(CONV3D_CASES))
def test_conv3d(golden, tag, case):
    g = golden("conv3d")
    k = f"{tag}_{case}_"
    kw = CONV3D_CASES[case]["kw"]
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi")]
    yr, yi = orc.cplx_conv3d(*a, g[k + "br"], g[k + "bi"], **kw)
    close(yr, g[k + "yr"], tag, 100)
    close(yi, g[k + "yi"], tag, 100)
    if kw.get("padding_mode", "zeros") == "zeros":
        bw = orc.cplx_conv3d_bwd(g[k + "gr"], g[k + "gi"], *a, **kw)
        for n in ("dxr", "dxi", "dwr", "dwi", "dbr", "dbi"):
            close(bw[n], g[k + n], tag, 300)


@pytest.mark.parametrize("tag", TAGS)
def test_lrt_conv3d(golden, tag):
    g = golden("conv3d")
    k = f"{tag}_vd_"
    kw = dict(stride=(1, 2, 1), padding=(1, 1, 0))
    er, ei = orc.cplx_randn_from_tape(g[k + "tape"])
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi")]
    yr, yi, _ = orc.lrt_cplx_conv3d(*a, g[k + "br"], g[k + "bi"], g[k + "ls2"], er, ei, **kw)
    close(yr, g[k + "yr"], tag, 100)
    close(yi, g[k + "yi"], tag, 100)
    mur, _ = orc.cplx_conv3d(*a, g[k + "br"], g[k + "bi"], **kw)
    close(mur, g[k + "yr_eval"], tag, 100)
    bw = orc.lrt_cplx_conv3d_bwd(g[k + "gr"], g[k + "gi"], *a, g[k + "ls2"], er, ei, **kw)
    for n, m in (("dxr", "dxr"), ("dxi", "dxi"), ("dwr", "dwr"), ("dwi", "dwi"), ("dbr", "dbr"),
                 ("dbi", "dbi"), ("dlog_sigma2", "dls2")):
        close(bw[n], g[k + m], tag, 1000)
    k = f"{tag}_real_"
    mu = orc.real_conv3d(g[k + "x"], g[k + "w"], padding=1) + g[k + "b"].reshape(-1, 1, 1, 1)
    close(mu, g[k + "y_eval"], tag, 100)
    s2 = orc.real_conv3d(g[k + "x"] ** 2, np.exp(g[k + "ls2"]), padding=1)
    close(mu + g[k + "eps"] * np.sqrt(np.maximum(s2, 1e-8)), g[k + "y"], tag, 100)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", list(POOL3D_CASES))
def test_oracle_max_pool3d(golden, tag, name):
    g = golden("conv3d")
    zr, zi = g[f"{tag}_mp_zr"], g[f"{tag}_mp_zi"]
    k = f"{tag}_mp_{name}_"
    yr, yi, idx = orc.cplx_max_pool3d(zr, zi, **POOL3D_CASES[name])
    assert np.array_equal(yr, g[k + "yr"]) and np.array_equal(yi, g[k + "yi"])
    B, C = zr.shape[:2]
    dzr = np.zeros((B, C, zr[0, 0].size), zr.dtype)
    for b in range(B):
        for c in range(C):
            np.add.at(dzr[b, c], idx[b, c].reshape(-1), g[k + "gr"][b, c].reshape(-1))
    close(dzr.reshape(zr.shape), g[k + "dzr"], tag, 10)


from oracle.gen_golden_cases import CONVT_CASES  # noqa: E402


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("case", list(CONVT_CASES))
def test_conv_transpose2d(golden, tag, case):
    g = golden("conv_transpose")
    k = f"{tag}_{case}_"
    kw = CONVT_CASES[case]["kw"]
    a = [g[k + n] for n in ("xr", "xi", "wr", "wi")]
    yr, yi = orc.cplx_conv_transpose2d(*a, g[k + "br"], g[k + "bi"], **kw)
    assert yr.shape == g[k + "yr"].shape
    close(yr, g[k + "yr"], tag, 100)
    close(yi, g[k + "yi"], tag, 100)
    bw = orc.cplx_conv_transpose2d_bwd(g[k + "gr"], g[k + "gi"], *a, **kw)
    for n in ("dxr", "dxi", "dwr", "dwi", "dbr", "dbi"):
        close(bw[n], g[k + n], tag, 300)


# ---- round 2 fixtures (tests/golden/r02.npz) ----------------------------------------------------
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_abs_and_log_alpha_gradients(golden, tag):
    g = golden("r02")
    tol = TOL[tag]
    zr, zi, up = g[f"{tag}_abs_zr"], g[f"{tag}_abs_zi"], g[f"{tag}_abs_g"]
    np.testing.assert_allclose(orc.cplx_abs(zr, zi), g[f"{tag}_abs_abs"], **tol)
    dzr, dzi = orc.cplx_abs_bwd(up, zr, zi)
    np.testing.assert_allclose(dzr, g[f"{tag}_abs_dzr"], **tol)
    np.testing.assert_allclose(dzi, g[f"{tag}_abs_dzi"], **tol)
    assert np.all(g[f"{tag}_abs_dzr"][(zr == 0) & (zi == 0)] == 0)       # the reference's subgradient at 0
    wr, wi, ls2, gs = (g[f"{tag}_sg_{k}"] for k in ("wr", "wi", "ls2", "g"))
    for kind in orc.KINDS:
        w_i = wi if kind.startswith("cplx") else None
        k = f"{tag}_sg_{kind}_"
        np.testing.assert_allclose(orc.log_alpha(ls2, wr, w_i), g[k + "la"], **tol)
        np.testing.assert_array_equal(g[k + "la_dls2"], gs)
        # the tiny-|w| entries amplify by 2/|w| ~ 1e20: compare relative to that scale
        dwr, dwi = orc.log_alpha_bwd(gs, wr, w_i)
        np.testing.assert_allclose(dwr, g[k + "la_dwr"], rtol=tol["rtol"] * 10, atol=tol["atol"])
        if w_i is not None:
            np.testing.assert_allclose(dwi, g[k + "la_dwi"], rtol=tol["rtol"] * 10, atol=tol["atol"])


@pytest.mark.parametrize("tag", ["f64"])
@pytest.mark.parametrize("kind", orc.KINDS)
def test_penalty_gradients_signed_cotangent(golden, tag, kind):
    """Negative / mixed-sign upstream gradients (the real kinds lost the sign)."""
    g = golden("r02")
    wr, wi, ls2, gs = (g[f"{tag}_sg_{k}"] for k in ("wr", "wi", "ls2", "g"))
    w_i = wi if kind.startswith("cplx") else None
    k = f"{tag}_sg_{kind}_"
    for pre, up in (("pen_", gs), ("negsum_", np.full_like(gs, -0.37))):
        o = orc.penalty_bwd(kind, up, ls2, wr, w_i)
        fin = np.isfinite(g[k + pre + "dls2"])
        np.testing.assert_allclose(o["dlog_sigma2"][fin], g[k + pre + "dls2"][fin], rtol=1e-7, atol=1e-9)
        fin = np.isfinite(g[k + pre + "dwr"])
        np.testing.assert_allclose(o["dwr"][fin], g[k + pre + "dwr"][fin], rtol=1e-6, atol=1e-9)
        if w_i is not None:
            fin = np.isfinite(g[k + pre + "dwi"])
            np.testing.assert_allclose(o["dwi"][fin], g[k + pre + "dwi"][fin], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_masked_layers_and_binarize(golden, tag):
    g = golden("r02")
    tol = TOL[tag]
    k = f"{tag}_mk_cl_"
    for m in ("hard", "soft"):
        mask = g[k + m + "_mask"]
        yr, yi = orc.cplx_linear(g[k + "xr"], g[k + "xi"], g[k + "wr"] * mask, g[k + "wi"] * mask, g[k + "br"], g[k + "bi"])
        np.testing.assert_allclose(yr, g[k + m + "_yr"], **tol)
        np.testing.assert_allclose(yi, g[k + m + "_yi"], **tol)
        bw = orc.cplx_linear_bwd(g[k + "gr"], g[k + "gi"], g[k + "xr"], g[k + "xi"], g[k + "wr"] * mask, g[k + "wi"] * mask)
        np.testing.assert_allclose(bw["dwr"] * mask, g[k + m + "_dwr"], **tol)
        np.testing.assert_allclose(bw["dwi"] * mask, g[k + m + "_dwi"], **tol)
        np.testing.assert_allclose(bw["dxr"], g[k + m + "_dxr"], **tol)
    assert list(g[k + "state_keys"]) == ["bias.imag", "bias.real", "mask", "weight.imag", "weight.real"]
    # binarize_masks incl. the -0.0 clean-up
    pre = f"{tag}_bz_"
    sd = {n[len(pre) + 3:]: v for n, v in g.items() if n.startswith(pre + "in_")}
    masks = {n[len(pre) + 9:]: v for n, v in g.items() if n.startswith(pre + "softmask_")}
    out, hard = orc.binarize_masks(sd, masks)
    for n, v in out.items():
        np.testing.assert_array_equal(v, g[pre + "out_" + n])
        np.testing.assert_array_equal(np.signbit(v), g[pre + "signbit_" + n])
    for n, v in hard.items():
        np.testing.assert_array_equal(v, g[pre + "hard_" + n])
    assert list(g[pre + "mask_keys"]) == ["a.mask", "b.mask"]
    assert list(g[pre + "named_masks"]) == ["a", "b"]


def test_philox_known_answers():
    """Random123's known-answer vectors for philox4x32 (kat_vectors: counter c0..c3, key k0 k1 -> output), with
    7 rounds (the kernels' stream, csrc/common.h kPhiloxRounds) and with 10 (Random123's default)."""
    from oracle import philox

    def run(ctr, key, rounds):
        group = np.array([ctr[0] | (ctr[1] << 32)], dtype=np.uint64)
        offset = ctr[2] | (ctr[3] << 32)
        seed = key[0] | (key[1] << 32)
        return [int(v) for v in philox.philox4x32(group, offset, seed, rounds)[0]]

    ones = [0xFFFFFFFF] * 4
    pi_c, pi_k = [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]
    assert run([0] * 4, [0, 0], 7) == [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]
    assert run(ones, [0xFFFFFFFF] * 2, 7) == [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]
    assert run(pi_c, pi_k, 7) == [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]
    assert run([0] * 4, [0, 0], 10) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run(ones, [0xFFFFFFFF] * 2, 10) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run(pi_c, pi_k, 10) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert philox.ROUNDS == 7
    # the normal stream built on it: moments of 2^18 draws
    z = philox.real_noise(1 << 18, 12345, 3)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01
