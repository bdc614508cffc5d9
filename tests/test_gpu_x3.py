"""float32-accurate products on the bf16 matrix pipe (cplxmodule_amd/x3.py, csrc/split.hip): the split itself, the three
launch sequences, and the float32 layers in 'x3' mode against the reference's float32 outputs (tests/golden/x3.npz,
lrt_linear.npz, linear.npz; oracle/gen_golden.py:gen_x3) AND against the float64 oracle on the same float32 inputs --
what the float32 numbers approximate -- at 2e-6 norm-wise."""
import numpy as np
import pytest
import torch

from oracle import cplx_oracle as orc

pytestmark = pytest.mark.gpu


def _norm_tol(ref, rtol):
    return dict(rtol=rtol, atol=rtol * float(np.abs(ref).max()))


def _bf(t):
    return t.float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("rows,cols", [(64, 96), (1, 8), (257, 40)])
def test_split3_exact_and_layouts(rows, cols):
    """x0 + x1 + x2 == x exactly (float32 arithmetic), pieces in the documented order, both layouts, both patterns."""
    from cplxmodule_amd import x3
    rs = np.random.RandomState(rows + cols)
    x = (rs.randn(rows, cols) * np.exp(rs.uniform(-20, 20, (rows, cols)))).astype(np.float32)
    x[0, 0], x[0, 1] = 0.0, -0.0
    t = torch.from_numpy(x).cuda()
    a = x3.split(t).t                                  # [rows, 3 cols] = [x2 | x1 | x0]
    p2, p1, p0 = (a[:, j * cols:(j + 1) * cols].float() for j in range(3))
    assert torch.equal(p0, t.bfloat16().float())
    assert torch.equal(p1, (t - p0).bfloat16().float())
    assert torch.equal((p2 + p1) + p0, t)              # exact reconstruction in float32
    assert float((p1.abs() > p0.abs() * 2.0 ** -7).sum()) == 0 and float((p2.abs() > p0.abs() * 2.0 ** -15).sum()) == 0
    b = x3.split(t, x3.SPLIT_B).t                      # [rows, 6 cols] = [x2 | x1 x1 | x0 x0 x0]
    for j, want in enumerate((p2, p1, p1, p0, p0, p0)):
        assert torch.equal(b[:, j * cols:(j + 1) * cols].float(), want)
    s = x3.split(t, x3.SPLIT_B, stacked=True).t        # [6, rows, cols]
    for j, want in enumerate((p2, p1, p1, p0, p0, p0)):
        assert torch.equal(s[j].float(), want)
    sa = x3.split(t, stacked=True).t
    assert torch.equal(sa[2].float(), p0) and torch.equal(sa[0].float(), p2)


def test_split3_ops_and_nonfinite():
    from cplxmodule_amd import ops, x3
    rs = np.random.RandomState(3)
    xr = torch.from_numpy(rs.randn(40, 64).astype(np.float32)).cuda()
    xi = torch.from_numpy(rs.randn(40, 64).astype(np.float32)).cuda()
    a = x3.split(xr, op=x3.OP_ABS2, t2=xi).t
    full = (a[:, :64].float() + a[:, 64:128].float()) + a[:, 128:].float()
    assert torch.equal(full, ops.abs2(xr, xi))         # the very arithmetic of the |x|^2 kernel, split exactly
    a = x3.split(xr, op=x3.OP_ABS2).t
    assert torch.equal((a[:, :64].float() + a[:, 64:128].float()) + a[:, 128:].float(), xr * xr)
    e = x3.split(xr, op=x3.OP_EXP).t
    assert torch.equal((e[:, :64].float() + e[:, 64:128].float()) + e[:, 128:].float(), ops.exp(xr))
    bad = xr.clone()
    bad[0, 0], bad[0, 1], bad[0, 2], bad[0, 3] = float("inf"), float("-inf"), float("nan"), 3.4e38
    p = x3.split(bad).t
    p2, p1, p0 = (p[:, j * 64:(j + 1) * 64].float() for j in range(3))
    assert p0[0, 0] == float("inf") and p0[0, 1] == float("-inf") and torch.isnan(p0[0, 2])
    assert float(p1[0, :3].abs().sum()) == 0 and float(p2[0, :3].abs().sum()) == 0
    assert torch.isfinite(p1[0, 3]) and torch.isfinite(p2[0, 3])        # a value that rounds to bf16 inf keeps its leading piece alone
    # strided source rows (a column block of a wider matrix) and refusal of shapes the kernel does not take
    wide = torch.from_numpy(rs.randn(16, 128).astype(np.float32)).cuda()
    v = wide[:, 32:96]
    pv = x3.split(v).t
    assert torch.equal((pv[:, :64].float() + pv[:, 64:128].float()) + pv[:, 128:].float(), v)
    from cplxmodule_amd._lib import CplxAmdError
    with pytest.raises(CplxAmdError):
        x3.split(torch.zeros(4, 12, device="cuda"))    # cols % 8 != 0


@pytest.mark.parametrize("M,N_,K", [(64, 96, 32), (256, 128, 192), (520, 264, 96), (512, 512, 1024)])
@pytest.mark.parametrize("cplx", (True, False))
@pytest.mark.parametrize("kind", ("x3", "x2"))
def test_x3_gemm_forms_vs_float64(M, N_, K, cplx, kind):
    """The three launch sequences (N,N) / (N,T) / (T,T) on split operands against float64 numpy: norm-wise 2e-6, and
    within a small factor of the exact float32-MFMA kernel on the same float32 inputs (both are float32 accumulations; the
    split products run three times the K depth in one accumulator chain)."""
    from cplxmodule_amd import ops, x3
    rs = np.random.RandomState(M + N_ + K)
    P = 2 if cplx else 1
    A = [rs.randn(M, K).astype(np.float32) for _ in range(P)]
    B = [(rs.randn(N_, K) * 0.1).astype(np.float32) for _ in range(P)]
    bias = [rs.randn(N_).astype(np.float32) for _ in range(P)]
    a64 = A[0].astype(np.float64) + (1j * A[1] if cplx else 0)
    b64 = B[0].astype(np.float64) + (1j * B[1] if cplx else 0)
    cu = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()  # noqa: E731
    At, Bt, bt = [cu(v) for v in A], [cu(v) for v in B], [cu(v) for v in bias]

    def check(got, ref, exact):
        got = got if isinstance(got, tuple) else (got,)
        exact = exact if isinstance(exact, tuple) else (exact,)
        for p, part in enumerate(("real", "imag")[:P]):
            r = getattr(ref, part) if cplx else ref
            scale = float(np.abs(r).max())
            e3 = float(np.abs(_bf(got[p]) - r).max()) / scale
            ee = float(np.abs(_bf(exact[p]) - r).max()) / scale
            np.testing.assert_allclose(_bf(got[p]), r, rtol=0, atol=2e-6 * scale)
            assert e3 <= max(4.0 * ee, 1e-6), (e3, ee)

    # (N,N): C = A B^T + bias
    ref = a64 @ b64.T + (bias[0] + (1j * bias[1] if cplx else 0))
    sp = lambda ts, *a, **k: x3.split_planes(tuple(ts), *a, kind=kind, **k)  # noqa: E731
    got = x3.gemm_nn(sp(At), sp(Bt, x3.SPLIT_B), M, N_, K, bias=tuple(bt) if cplx else bt[0])
    if cplx:
        exact = ops.cgemm(At[0], At[1], (K, 1), Bt[0], Bt[1], (K, 1), M, N_, K, bias=tuple(bt))
    else:
        exact = ops.rgemm(At[0], (K, 1), Bt[0], (K, 1), M, N_, K, bias=bt[0])
    check(got, ref, exact)
    # (N,T): C[m, n] = sum_k A[m, k] conj(Bk[k, n]) with Bk = B^T stored [K, N]
    Bk = [cu(v.T.copy()) for v in B]
    ref = a64 @ (b64.conj() if cplx else b64).T
    got = x3.gemm_nt(sp(At), sp(Bk, x3.SPLIT_B, stacked=True), M, N_, K, conj_b=cplx)
    if cplx:
        exact = ops.cgemm(At[0], At[1], (K, 1), Bk[0], Bk[1], (1, N_), M, N_, K, conj_b=True)
    else:
        exact = ops.rgemm(At[0], (K, 1), Bk[0], (1, N_), M, N_, K)
    check(got, ref, exact)
    # (T,T): C[m, n] = sum_k Ak[k, m] conj(Bk[k, n]), accumulated on top of beta * C0
    Ak = [cu(v.T.copy()) for v in A]
    C0 = [rs.randn(M, N_).astype(np.float32) for _ in range(P)]
    beta = torch.tensor(0.25, device="cuda")
    ref = a64 @ (b64.conj() if cplx else b64).T + 0.25 * (C0[0] + (1j * C0[1] if cplx else 0))
    out = tuple(cu(v) for v in C0) if cplx else cu(C0[0])
    got = x3.gemm_tt(sp(Ak), sp(Bk), M, N_, K, conj_b=cplx, out=out, accumulate=True, beta=beta)
    out2 = tuple(cu(v) for v in C0) if cplx else cu(C0[0])
    if cplx:
        exact = ops.cgemm(Ak[0], Ak[1], (1, M), Bk[0], Bk[1], (1, N_), M, N_, K, conj_b=True, out=out2, accumulate=True,
                          beta=beta)
    else:
        exact = ops.rgemm(Ak[0], (1, M), Bk[0], (1, N_), M, N_, K, out=out2, accumulate=True, beta=beta)
    check(got, ref, exact)


def test_take_rules():
    from cplxmodule_amd import x3
    t = torch.zeros(8, device="cuda")
    assert x3.take(64, 64, 32, t, mode="x3") == "x3" and x3.take(64, 64, 32, t, mode="x2") == "x2"
    assert not x3.take(64, 64, 32, t, mode="exact")
    assert not x3.take(64, 64, 32, t, mode="auto") and x3.take(1024, 1024, 1024, t, mode="auto")
    assert not x3.take(64, 60, 32, t, mode="x3") and not x3.take(64, 64, 48, t, mode="x3")
    assert not x3.take(64, 64, 32, t.bfloat16(), mode="x3") and not x3.take(1 << 21, 64, 32, t, mode="x3")
    with x3.fp32_mode("x3"):
        assert x3.get_fp32_mode() == "x3"
    with pytest.raises(ValueError):
        x3.set_fp32_mode("fast")


def _count_splits(monkeypatch):
    from cplxmodule_amd import x3
    n = {"split": 0}
    orig = x3.split

    def counting(*a, **k):
        n["split"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(x3, "split", counting)
    return n


@pytest.mark.parametrize("mode", ("x3", "x2"))
def test_cplx_linear_x3_golden(golden, monkeypatch, mode):
    """cplx.linear in x3 mode vs the reference: float32 outputs / autograd gradients at the suite's tolerance, the float64
    ones (same seeds) at 2e-6 norm-wise."""
    from gpu_util import T, N
    from cplxmodule_amd import cplx, x3
    g = golden("x3")
    n = _count_splits(monkeypatch)
    leaves = {m: T(g["f32_lin_" + m]).requires_grad_(True) for m in ("xr", "xi", "wr", "wi", "br", "bi")}
    with x3.fp32_mode(mode):
        y = cplx.linear(cplx.Cplx(leaves["xr"], leaves["xi"]), cplx.Cplx(leaves["wr"], leaves["wi"]),
                        cplx.Cplx(leaves["br"], leaves["bi"]))
    assert n["split"] == 4                               # x (2 planes) + W (2 planes): the split path ran
    ((y.real * T(g["f32_lin_gr"])).sum() + (y.imag * T(g["f32_lin_gi"])).sum()).backward()
    assert n["split"] == 4 + 2 + 2                       # G once (shared by dW and dX) + W stacked; the pieces of x were kept
    got = dict(yr=y.real, yi=y.imag, dxr=leaves["xr"].grad, dxi=leaves["xi"].grad, dwr=leaves["wr"].grad,
               dwi=leaves["wi"].grad, dbr=leaves["br"].grad, dbi=leaves["bi"].grad)
    f = {m: g["f32_lin_" + m].astype(np.float64) for m in ("xr", "xi", "wr", "wi", "br", "bi", "gr", "gi")}
    t64 = dict(zip(("yr", "yi"), orc.cplx_linear(f["xr"], f["xi"], f["wr"], f["wi"], f["br"], f["bi"])))
    t64.update(orc.cplx_linear_bwd(f["gr"], f["gi"], f["xr"], f["xi"], f["wr"], f["wi"]))
    for m, t in got.items():
        np.testing.assert_allclose(N(t), g["f32_lin_" + m], **_norm_tol(g["f32_lin_" + m], 1e-5), err_msg=m)
        np.testing.assert_allclose(N(t), t64[m], **_norm_tol(t64[m], 2e-6), err_msg=m + " (float64 oracle)")


@pytest.mark.parametrize("case", "abc")
def test_cplx_linear_fp32_golden_under_x3(golden, case):
    """The suite's own golden test with x3 forced: case b takes the split products, a and c (K = 200, 33) stay exact."""
    import test_gpu_linear as tl
    import cplxmodule_amd
    from cplxmodule_amd import x3
    with x3.fp32_mode("x3"):
        tl.test_cplx_linear_fp32_golden(golden, cplxmodule_amd, case)


def _cvd_layer(g, k, O, I):
    from gpu_util import T
    from cplxmodule_amd.nn import relevance as rel
    layer = rel.CplxLinearVD(I, O).to("cuda")
    layer.load_state_dict({"weight.real": T(g[k + "wr"]), "weight.imag": T(g[k + "wi"]), "bias.real": T(g[k + "br"]),
                           "bias.imag": T(g[k + "bi"]), "log_sigma2": T(g[k + "ls2"])})
    return layer


@pytest.mark.parametrize("mode", ("x3", "x2", "exact"))
def test_cplx_linear_vd_x3_golden(golden, monkeypatch, mode):
    """CplxLinearVD (training mode, the reference's noise tape, loss + 1e-2 KL) in x3 mode -- and, for the comparison in
    the parity report, in exact mode -- vs the reference's float32 and float64 numbers."""
    from gpu_util import T, N
    from cplxmodule_amd import cplx, x3
    from cplxmodule_amd.nn import relevance as rel
    g = golden("x3")
    k = "f32_cvd_"
    O, I = g[k + "wr"].shape
    n = _count_splits(monkeypatch)
    layer = _cvd_layer(g, k, O, I)
    xr, xi = T(g[k + "xr"]).requires_grad_(True), T(g[k + "xi"]).requires_grad_(True)
    tape = T(g[k + "tape"]) / np.float32(np.sqrt(2.0))
    layer.train()
    with x3.fp32_mode(mode):
        y = layer(cplx.Cplx(xr, xi), eps=cplx.Cplx(tape[0], tape[1]))
        kl = sum(rel.penalties(layer))
    assert (n["split"] > 0) == (mode != "exact")
    np.testing.assert_allclose(float(kl), float(g[k + "kl"]), rtol=1e-5)
    ((y.real * T(g[k + "gr"])).sum() + (y.imag * T(g[k + "gi"])).sum() + 1e-2 * kl).backward()
    got = dict(yr=y.real, yi=y.imag, dxr=xr.grad, dxi=xi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad,
               dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad, dls2=layer.log_sigma2.grad)
    f = {m: g[k + m].astype(np.float64) for m in ("xr", "xi", "wr", "wi", "br", "bi", "ls2", "gr", "gi")}
    er, ei = orc.cplx_randn_from_tape(g[k + "tape"].astype(np.float64))
    yr, yi, _ = orc.lrt_cplx_linear(f["xr"], f["xi"], f["wr"], f["wi"], f["br"], f["bi"], f["ls2"], er, ei)
    t64 = orc.lrt_cplx_linear_bwd(f["gr"], f["gi"], f["xr"], f["xi"], f["wr"], f["wi"], f["ls2"], er, ei)
    klg = orc.penalty_bwd("cplx_vd", np.full_like(f["ls2"], 1e-2), f["ls2"], f["wr"], f["wi"])
    t64.update(yr=yr, yi=yi, dwr=t64["dwr"] + klg["dwr"], dwi=t64["dwi"] + klg["dwi"],
               dls2=t64["dlog_sigma2"] + klg["dlog_sigma2"])
    for m, t in got.items():
        np.testing.assert_allclose(N(t), g[k + m], **_norm_tol(g[k + m], 1e-5), err_msg=m)
        np.testing.assert_allclose(N(t), t64[m], **_norm_tol(t64[m], 2e-6), err_msg=m + " (float64 oracle)")


@pytest.mark.parametrize("mode", ("x3", "x2"))
def test_real_linear_vd_x3_golden(golden, monkeypatch, mode):
    from gpu_util import T, N
    from cplxmodule_amd import x3
    from cplxmodule_amd.nn import relevance as rel
    g = golden("x3")
    k = "f32_rvd_"
    O, I = g[k + "w"].shape
    n = _count_splits(monkeypatch)
    layer = rel.LinearVD(I, O).to("cuda")
    layer.load_state_dict({"weight": T(g[k + "w"]), "bias": T(g[k + "b"]), "log_sigma2": T(g[k + "ls2"])})
    x = T(g[k + "x"]).requires_grad_(True)
    layer.train()
    with x3.fp32_mode(mode):
        y = layer(x, eps=T(g[k + "eps"]))
        kl = sum(rel.penalties(layer))
    assert n["split"] > 0
    ((y * T(g[k + "g"])).sum() + 1e-2 * kl).backward()
    f = {m: g[k + m].astype(np.float64) for m in ("x", "w", "b", "ls2", "g", "eps")}
    t64 = orc.lrt_real_linear_bwd(f["g"], f["x"], f["w"], f["ls2"], f["eps"])
    klg = orc.penalty_bwd("real_vd", np.full_like(f["ls2"], 1e-2), f["ls2"], f["w"])
    t64.update(y=orc.lrt_real_linear(f["x"], f["w"], f["b"], f["ls2"], f["eps"])[0], dw=t64["dw"] + klg["dwr"],
               dls2=t64["dlog_sigma2"] + klg["dlog_sigma2"])
    for m, t in dict(y=y, dx=x.grad, dw=layer.weight.grad, db=layer.bias.grad, dls2=layer.log_sigma2.grad).items():
        np.testing.assert_allclose(N(t), g[k + m], **_norm_tol(g[k + m], 1e-5), err_msg=m)
        np.testing.assert_allclose(N(t), t64[m], **_norm_tol(t64[m], 2e-6), err_msg=m + " (float64 oracle)")


@pytest.mark.parametrize("mode", ("x3", "x2"))
def test_lrt_goldens_under_x3(golden, mode):
    """cfg1's shapes (B = 64, 128 -> 128): the suite's LRT golden tests (complex, clamp boundary, real) with the split
    arithmetic forced."""
    import test_gpu_linear as tl
    import cplxmodule_amd
    from cplxmodule_amd import x3
    with x3.fp32_mode(mode):
        tl.test_lrt_cplx_linear_layer_golden(golden, cplxmodule_amd)
        tl.test_lrt_clamp_boundary_layer(golden, cplxmodule_amd)
        tl.test_lrt_real_linear_layer_golden(golden, cplxmodule_amd)


def test_x3_ragged_batch_falls_back_per_product(monkeypatch):
    """B = 40 (not a multiple of 32): forward and input gradient take the split products, the weight gradients (K = B)
    the exact kernel -- one layer, both arithmetics, same answers as exact mode to 2e-6."""
    from cplxmodule_amd import cplx, x3
    from cplxmodule_amd.nn import relevance as rel
    torch.manual_seed(5)
    layer = rel.CplxLinearVD(64, 96).to("cuda")
    with torch.no_grad():
        layer.log_sigma2.uniform_(-8, 0)
    x = cplx.Cplx(torch.randn(40, 64, device="cuda"), torch.randn(40, 64, device="cuda"))
    eps = cplx.Cplx(torch.randn(40, 96, device="cuda"), torch.randn(40, 96, device="cuda"))
    res = {}
    for mode in ("exact", "x3", "x2"):
        layer.zero_grad()
        xr, xi = x.real.clone().requires_grad_(True), x.imag.clone().requires_grad_(True)
        with x3.fp32_mode(mode):
            y = layer(cplx.Cplx(xr, xi), eps=eps)
        (y.real.square().sum() + y.imag.square().sum()).backward()
        res[mode] = [t.detach().cpu().double().numpy() for t in (y.real, y.imag, xr.grad, xi.grad, layer.weight.real.grad,
                                                                layer.weight.imag.grad, layer.log_sigma2.grad)]
    for m in ("x3", "x2"):
        for a, b in zip(res[m], res["exact"]):
            np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * float(np.abs(b).max()))


def test_x3_masked_layer():
    """Masked layer: weight * mask before the split, dW * mask in every term of the six-launch weight gradient."""
    from cplxmodule_amd import cplx, x3
    from cplxmodule_amd.nn import masked
    torch.manual_seed(7)
    layer = masked.CplxLinearMasked(64, 96).to("cuda")
    mask = (torch.rand(96, 64, device="cuda") > 0.4).float()
    layer.mask = mask
    x = cplx.Cplx(torch.randn(64, 64, device="cuda"), torch.randn(64, 64, device="cuda"))
    res = {}
    for mode in ("exact", "x3"):
        layer.zero_grad()
        with x3.fp32_mode(mode):
            y = layer(x)
        (y.real.square().sum() + y.imag.square().sum()).backward()
        res[mode] = [t.detach().cpu().double().numpy() for t in (y.real, y.imag, layer.weight.real.grad, layer.weight.imag.grad)]
    for a, b in zip(res["x3"], res["exact"]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * float(np.abs(b).max()))
    assert float((torch.from_numpy(res["x3"][2]).cuda().float() * (1 - mask)).abs().max()) == 0.0


@pytest.mark.parametrize("scale_pow", (0, -30, 20))
def test_split2h_pieces_and_scale(scale_pow):
    """IEEE-half pieces: scale = the power of two that puts max |x| into [2^14, 2^15); x s = h0 + h1 + r with
    |r| <= 2^-22 |x s| for every element within 2^17 of the largest, <= 2^-25 (scaled units) below; layouts as the bf16
    pieces; the planes of a complex operand share one scale."""
    from cplxmodule_amd import x3
    rs = np.random.RandomState(11)
    x = (rs.randn(96, 64) * np.exp(rs.uniform(-12, 0, (96, 64))) * 2.0 ** scale_pow).astype(np.float32)
    x[0, 0] = 0.0
    t = torch.from_numpy(x).cuda()
    sc = x3.scale_of(t)
    s_, inv = float(sc[0]), float(sc[1])
    amax = float(np.abs(x).max())
    assert s_ * inv == 1.0 and 2.0 ** 14 <= amax * s_ < 2.0 ** 15 and np.log2(s_) == round(np.log2(s_))
    p = x3.split(t, kind="x2")
    assert p.t.dtype == torch.float16 and p.n == 2 and p.t.shape == (96, 128) and torch.equal(p.scale, sc)
    h1, h0 = p.t[:, :64].double(), p.t[:, 64:].double()
    xs = t.double() * s_
    assert torch.equal(h0.half(), (t * s_).half().to(h0.dtype).half())          # h0 = half(x s)
    r = (xs - h0 - h1).abs()
    bound = torch.maximum(xs.abs() * 2.0 ** -22, torch.full_like(xs, 2.0 ** -25))
    assert bool((r <= bound).all())
    b = x3.split(t, x3.SPLIT_B, kind="x2", scale=sc)
    assert b.n == 3 and torch.equal(b.t[:, :64], p.t[:, :64]) and torch.equal(b.t[:, 64:128], p.t[:, 64:])
    assert torch.equal(b.t[:, 128:], p.t[:, 64:])
    st = x3.split(t, x3.SPLIT_B, kind="x2", scale=sc, stacked=True)
    assert torch.equal(st.t[0], p.t[:, :64]) and torch.equal(st.t[2], p.t[:, 64:])
    # complex pair: one scale from the larger plane; all-zero tensor: scale 1
    y = torch.from_numpy((rs.randn(96, 64) * 37.0).astype(np.float32)).cuda()
    pr, pi = x3.split_planes((t, y), kind="x2")
    assert pr.scale is pi.scale and 2.0 ** 14 <= max(amax, float(y.abs().max())) * float(pr.scale[0]) < 2.0 ** 15
    z = x3.scale_of(torch.zeros(8, 8, device="cuda"))
    assert float(z[0]) == 1.0 and float(z[1]) == 1.0
    a2 = x3.split(t, op=x3.OP_ABS2, t2=y, kind="x2")
    full = (a2.t[:, :64].double() + a2.t[:, 64:].double()) * float(a2.scale[1])
    ref = t.double() ** 2 + y.double() ** 2
    assert float((full - ref).abs().max()) <= 2.0 ** -21 * float(ref.max())


def test_x2_dynamic_range_is_normwise_accurate():
    """Operands spanning 30 binades (gradient-like magnitudes 1e-9 .. 1e-3 against weights of 0.05): the half split with
    per-operand scales stays at 2^-22 norm-wise -- and the bf16 split at 2^-24 -- against float64."""
    from cplxmodule_amd import x3
    rs = np.random.RandomState(5)
    M, N_, K = 256, 128, 512
    A = (rs.randn(M, K) * np.exp(rs.uniform(-21, -7, (M, K)))).astype(np.float32)
    B = (rs.randn(N_, K) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    At, Bt = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    for kind, tol in (("x2", 1e-6), ("x3", 4e-7)):
        got = x3.gemm_nn(x3.split_planes((At,), kind=kind), x3.split_planes((Bt,), x3.SPLIT_B, kind=kind), M, N_, K)
        err = float(np.abs(got.double().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
        assert err <= tol, (kind, err)


# ---- float32 3 x 3 convolutions on IEEE-half pieces (conv.py: _x2_conv*, csrc/conv_cl2_f16.hip, conv_cl_wgrad_f16.hip) -------
@pytest.mark.parametrize("pad,H,W,bias", [(1, 20, 40, True), (0, 35, 33, False), (1, 16, 32, True)])
def test_x2_conv_vs_float64_and_exact(pad, H, W, bias):
    """CplxConv2d(64, 128, 3) float32, channels-last and planar inputs: forward, input gradient, weight and bias gradients
    in 'x2' mode and in exact mode against the float64 oracle (cplx.py:717-838 restated) at 4e-6 norm-wise (K = 9 x 128
    float32 accumulations: the exact kernel itself reaches 2.2e-6 here)."""
    from cplxmodule_amd import Cplx, conv, nn, x3
    rs = np.random.RandomState(H + W + pad)
    B, Ci, Co = 3, 64, 128
    layer = nn.CplxConv2d(Ci, Co, 3, padding=pad, bias=bias).cuda()
    xr, xi = (rs.randn(B, Ci, H, W).astype(np.float32) for _ in range(2))
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    gr, gi = (rs.randn(B, Co, Ho, Wo).astype(np.float32) for _ in range(2))
    f = np.float64
    wr, wi = (layer.weight.real.detach().cpu().numpy().astype(f), layer.weight.imag.detach().cpu().numpy().astype(f))
    br = bi = None
    if bias:
        br, bi = layer.bias.real.detach().cpu().numpy().astype(f), layer.bias.imag.detach().cpu().numpy().astype(f)
    yr64, yi64 = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr, wi, br, bi, padding=pad)
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr, wi, padding=pad)
    res = {}
    for mode, cl in (("exact", True), ("x2", True), ("x2", False)):
        layer.zero_grad()
        mk = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
        txr, txi = mk(xr), mk(xi)
        if cl:
            txr, txi = (t.contiguous(memory_format=torch.channels_last) for t in (txr, txi))
        txr.requires_grad_(True); txi.requires_grad_(True)
        with x3.fp32_mode(mode):
            assert (conv._x2_conv_kind(conv._geom(txr.shape, layer.weight.real.shape, 1, pad, 1, 1)[0], txr, txi,
                                       layer.weight.real, layer.weight.imag) == "x2") == (mode == "x2")
            y = layer(Cplx(txr, txi))
        torch.autograd.backward((y.real, y.imag), (mk(gr), mk(gi)))
        got = dict(yr=y.real, yi=y.imag, dxr=txr.grad, dxi=txi.grad, dwr=layer.weight.real.grad, dwi=layer.weight.imag.grad)
        if bias:
            got.update(dbr=layer.bias.real.grad, dbi=layer.bias.imag.grad)
        res[(mode, cl)] = {k: v.detach().double().cpu().numpy() for k, v in got.items()}
    ref = dict(yr=yr64, yi=yi64, dxr=bw["dxr"], dxi=bw["dxi"], dwr=bw["dwr"], dwi=bw["dwi"])
    if bias:
        ref.update(dbr=gr.astype(f).sum((0, 2, 3)), dbi=gi.astype(f).sum((0, 2, 3)))
    for key, got in res.items():
        for k, r in ref.items():
            np.testing.assert_allclose(got[k], r, rtol=0, atol=4e-6 * float(np.abs(r).max()), err_msg=f"{key} {k}")
    assert res[("x2", True)]["yr"].shape == (B, Co, Ho, Wo)


def test_x2_conv_batch_chunks(monkeypatch):
    """The 4-GB buffer-descriptor limit chunks the batch: force chunks of two images and compare with one launch set."""
    from cplxmodule_amd import Cplx, conv, nn, x3
    torch.manual_seed(8)
    layer = nn.CplxConv2d(64, 64, 3, padding=1).cuda()
    x = Cplx(*(torch.randn(5, 64, 24, 32, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
               for _ in range(2)))
    out = {}
    for tag, lim in (("one", conv._X2_BYTES_MAX), ("chunks", 2 * 24 * 32 * 128 * 2 + 1)):
        monkeypatch.setattr(conv, "_X2_BYTES_MAX", lim)
        layer.zero_grad(); x.real.grad = x.imag.grad = None
        with x3.fp32_mode("x2"):
            y = layer(x)
        torch.autograd.backward((y.real, y.imag), (y.real.detach(), y.imag.detach()))
        out[tag] = [t.detach().clone() for t in (y.real, y.imag, x.real.grad, x.imag.grad, layer.weight.real.grad)]
    for a, b in zip(out["one"][:4], out["chunks"][:4]):
        assert torch.equal(a, b)                              # forward / data gradient: per-image results, same bits
    a, b = out["one"][4], out["chunks"][4]
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())      # weight gradient: a sum over chunks


def test_x2_conv_backward_takes_the_scale_from_the_batchnorm_backward():
    """float32 conv (half pieces) -> batch-norm: the layer's backward apply leaves the power-of-two scale of max |dX| on the
    planes it writes (cplxamd_bn_bwd_sums_amax + cplxamd_absmax_scale_partials); it equals the scale the absmax pass computes
    from the planes, the convolution's backward uses it, and the gradients are the same bits as without the hint."""
    from cplxmodule_amd import Cplx, fp32_mode, nn, ops, x3
    dev = "cuda"
    res, seen = {}, []
    real_hint = ops.scale_hint
    for use in (True, False):
        ops.scale_hint = (lambda a, b: (seen.append(real_hint(a, b)), seen[-1])[1]) if use else (lambda a, b: None)
        try:
            torch.manual_seed(0)
            layer, bn = nn.CplxConv2d(64, 64, 3, padding=1).to(dev), nn.CplxBatchNorm2d(64).to(dev)
            mk = lambda: (torch.randn(4, 64, 48, 64, device=dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
                          .requires_grad_(True))
            x = Cplx(mk(), mk())
            with fp32_mode("x2"):
                y = bn(layer(x))
                g = (torch.randn_like(y.real), torch.randn_like(y.imag))
                torch.autograd.backward((y.real, y.imag), g)
            res[use] = [x.real.grad, x.imag.grad, layer.weight.real.grad, layer.weight.imag.grad]
        finally:
            ops.scale_hint = real_hint
    assert len(seen) == 1 and seen[0] is not None
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    # the hinted scale against the absmax pass on an arbitrary pair of planes
    from cplxmodule_amd._lib import call, ptr, stream_ptr
    p = torch.randn(5000, 64, device=dev) * 37.0
    q = torch.randn(5000, 64, device=dev)
    ref = x3.scale_of(p, q, x3.OP_MAX2)
    part = torch.zeros(2048, device=dev)
    part[:7] = torch.stack([t.abs().max() for t in (p[:900], q, p[900:], q[:3], p[:1], q[:1], p[4000:])])
    out = torch.empty(2, device=dev)
    call("cplxamd_absmax_scale_partials", ptr(part), 2048, ptr(out), stream_ptr())
    assert torch.equal(ref, out)
