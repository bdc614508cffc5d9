"""BASELINE.json's FULL sizes, checked through size-independent properties (the oracle cannot run
these in seconds): exact scaling / conjugation / row-subset / translation identities, sampled
rows against the float64 oracle, whitening and noise statistics, exact KL on the full weight.

  cfg2  CplxLinear 4096 -> 4096, bf16, batch 8192            (forward, dX, dW)
  cfg4  CplxLinearVD 2048 -> 2048 LRT + KL, bf16, batch 2^20
  cfg3  CplxConv2d(64, 64, 3) on 256 x 256 + CplxBatchNorm2d, bf16, batch 256
"""
import numpy as np
import pytest
import torch

import oracle.cplx_oracle as orc
from gpu_util import DEV, N

pytestmark = pytest.mark.gpu


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).bfloat16()


def test_cfg2_linear_full_size_properties():
    from cplxmodule_amd import Cplx, cplx
    B, I, O = 8192, 4096, 4096
    xr, xi = _bf(B, I, seed=1), _bf(B, I, seed=2)
    wr, wi = _bf(O, I, scale=0.02, seed=3).float(), _bf(O, I, scale=0.02, seed=4).float()
    br, bi = _bf(O, seed=5).float(), _bf(O, seed=6).float()
    x, w, b = Cplx(xr, xi), Cplx(wr, wi), Cplx(br, bi)
    y = cplx.linear(x, w, b)
    assert y.real.shape == (B, O) and torch.isfinite(y.real.float()).all()
    # (1) scaling by a power of two commutes bit for bit (no bias): 2 f(x) == f(2 x)
    y0, y2 = cplx.linear(x, w), cplx.linear(Cplx(xr * 2, xi * 2), w)
    assert torch.equal(y2.real, y0.real * 2) and torch.equal(y2.imag, y0.imag * 2)
    # (2) conjugation: f(conj x; conj W, conj b) == conj f(x; W, b).  The real part repeats the same
    # products bit for bit; the imaginary part negates every product, and the MFMA accumulation is
    # not sign-symmetric in its last bit, so it may differ by one bf16 ulp on a few elements
    yc = cplx.linear(Cplx(xr, -xi), Cplx(wr, -wi), Cplx(br, -bi))
    assert torch.equal(yc.real, y.real)
    d = (yc.imag.float() + y.imag.float()).abs()
    assert float((d / y.imag.float().abs().clamp_min(1e-3)).max()) <= 2 ** -7
    assert float((d > 0).float().mean()) < 0.02
    # (3) any subset of rows gives the same rows (tiles do not leak into each other)
    rows = torch.tensor([0, 1, 255, 256, 257, 4095, 4096, 8191], device=DEV)
    ys = cplx.linear(Cplx(xr[rows].contiguous(), xi[rows].contiguous()), w, b)
    assert torch.equal(ys.real, y.real[rows]) and torch.equal(ys.imag, y.imag[rows])
    # (4) the same rows against the float64 oracle (bf16 output rounding: 2^-8 relative to |y|)
    f = np.float64
    wr16, wi16 = N(wr.bfloat16().float()).astype(f), N(wi.bfloat16().float()).astype(f)
    rr, ri = orc.cplx_linear(N(xr[rows].float()).astype(f), N(xi[rows].float()).astype(f), wr16, wi16,
                             N(br).astype(f), N(bi).astype(f))
    scale = np.abs(rr).max()
    np.testing.assert_allclose(N(ys.real.float()), rr, rtol=8e-3, atol=8e-3 * scale)
    np.testing.assert_allclose(N(ys.imag.float()), ri, rtol=8e-3, atol=8e-3 * scale)
    # (5) backward: dX rows vs oracle, dW linear in the upstream gradient (exactly, for a factor 2)
    gr, gi = _bf(B, O, seed=7), _bf(B, O, seed=8)

    def grads(scale_):
        leaves = [t.clone().requires_grad_(True) for t in (xr, xi, wr, wi)]
        out = cplx.linear(Cplx(leaves[0], leaves[1]), Cplx(leaves[2], leaves[3]))
        torch.autograd.backward((out.real, out.imag), (gr * scale_, gi * scale_))
        return [t.grad for t in leaves]
    g1, g2 = grads(1.0), grads(2.0)
    for a, b_ in zip(g1, g2):
        assert torch.equal(b_, a * 2)
    bw = orc.cplx_linear_bwd(N(gr[rows].float()).astype(f), N(gi[rows].float()).astype(f),
                             N(xr[rows].float()).astype(f), N(xi[rows].float()).astype(f), wr16, wi16, has_bias=False)
    sc = np.abs(bw["dxr"]).max()
    np.testing.assert_allclose(N(g1[0][rows].float()), bw["dxr"], rtol=8e-3, atol=8e-3 * sc)
    np.testing.assert_allclose(N(g1[1][rows].float()), bw["dxi"], rtol=8e-3, atol=8e-3 * sc)
    # dW = G^T conj(X) on a sampled block of output rows, float64 over the full batch
    o_rows = [0, 777, 4095]
    G = N(gr[:, o_rows].float()).astype(f) + 1j * N(gi[:, o_rows].float()).astype(f)
    X = N(xr.float()).astype(f) + 1j * N(xi.float()).astype(f)
    dW = G.T @ X.conj()
    sw = np.abs(dW).max()
    np.testing.assert_allclose(N(g1[2][o_rows]), dW.real, rtol=1e-4, atol=2e-5 * sw)
    np.testing.assert_allclose(N(g1[3][o_rows]), dW.imag, rtol=1e-4, atol=2e-5 * sw)


def test_cfg4_lrt_kl_full_size_properties():
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    B, F = 1 << 20, 2048
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(F, F).to(DEV)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-12, 4)
    # exact KL of the FULL weight against the float64 oracle (scipy Ei on 4 M elements)
    kl = sum(rel.penalties(layer, reduction="sum"))
    w = layer.weight
    ref = orc.penalty("cplx_vd", N(layer.log_sigma2).astype(np.float64), N(w.real).astype(np.float64),
                      N(w.imag).astype(np.float64)).sum()
    np.testing.assert_allclose(float(kl.detach()), ref, rtol=2e-6)
    x = Cplx(_bf(B, F, seed=11), _bf(B, F, seed=12))
    layer.eval()
    mu = layer(x)                                   # eval: the mean path only
    layer.train()
    rel.noise.manual_seed(5)
    y1 = layer(x)
    rel.noise.manual_seed(5)
    y2 = layer(x)
    assert torch.equal(y1.real, y2.real) and torch.equal(y1.imag, y2.imag)       # counter-based noise
    y3 = layer(x)
    assert not torch.equal(y1.real, y3.real)                                      # next offset: fresh noise
    # noise statistics over the full 2^31 outputs: (y - mu) / sd has mean 0 and E|.|^2 = 1
    rows = torch.arange(0, B, 4099, device=DEV)
    a = (x.real[rows].float() ** 2 + x.imag[rows].float() ** 2).bfloat16().float()
    S = layer.log_sigma2.exp().bfloat16().float()
    sd = (a @ S.t()).clamp_min(1e-8).sqrt()
    zr = (y1.real[rows].float() - mu.real[rows].float()) / sd
    zi = (y1.imag[rows].float() - mu.imag[rows].float()) / sd
    n = zr.numel()
    assert abs(float(zr.mean())) < 5 / np.sqrt(n) + 2e-3 and abs(float(zi.mean())) < 5 / np.sqrt(n) + 2e-3
    assert abs(float((zr ** 2 + zi ** 2).mean()) - 1.0) < 2e-2                   # bf16 rounding of y, mu
    assert abs(float((zr * zi).mean())) < 2e-2
    # sampled rows of the mean path against the float64 oracle
    f = np.float64
    few = rows[:8]
    rr, ri = orc.cplx_linear(N(x.real[few].float()).astype(f), N(x.imag[few].float()).astype(f),
                             N(w.real.bfloat16().float()).astype(f), N(w.imag.bfloat16().float()).astype(f),
                             N(layer.bias.real).astype(f), N(layer.bias.imag).astype(f))
    np.testing.assert_allclose(N(mu.real[few].float()), rr, rtol=8e-3, atol=8e-3 * np.abs(rr).max())
    np.testing.assert_allclose(N(mu.imag[few].float()), ri, rtol=8e-3, atol=8e-3 * np.abs(ri).max())
    # backward at full size: finite, and dlog_sigma2 scales exactly with the upstream gradient
    del mu, y2, y3

    def dls2(scale_):
        layer.zero_grad(set_to_none=True)
        rel.noise.manual_seed(9)
        y = layer(x)
        torch.autograd.backward((y.real, y.imag), (y.real.detach() * scale_, y.imag.detach() * scale_))
        return layer.log_sigma2.grad.clone(), layer.weight.real.grad.clone()
    a1, w1 = dls2(1.0)
    a2, w2 = dls2(2.0)
    assert torch.isfinite(a1).all() and torch.isfinite(w1).all()
    assert torch.equal(a2, a1 * 2) and torch.equal(w2, w1 * 2)


def test_cfg3_conv_bn_full_size_properties():
    from cplxmodule_amd import Cplx, cplx, nn
    B, C, H = 256, 64, 256
    xr, xi = _bf(B, C, H, H, seed=21), _bf(B, C, H, H, seed=22)
    torch.manual_seed(1)
    conv = nn.CplxConv2d(C, C, 3).to(DEV)
    w, b = conv.weight, conv.bias
    y = conv(Cplx(xr, xi))
    assert y.real.shape == (B, C, H - 2, H - 2)
    # (1) exact scaling (no bias) and conjugation symmetry
    y0 = cplx.conv2d(Cplx(xr, xi), w)
    y2 = cplx.conv2d(Cplx(xr * 2, xi * 2), w)
    assert torch.equal(y2.real, y0.real * 2) and torch.equal(y2.imag, y0.imag * 2)
    del y2
    yc = cplx.conv2d(Cplx(xr, -xi), Cplx(w.real, -w.imag), Cplx(b.real, -b.imag))
    assert torch.equal(yc.real, y.real)
    d = (yc.imag.float() + y.imag.float()).abs()             # see the note in the cfg2 test
    assert float((d / y.imag.float().abs().clamp_min(1e-3)).max()) <= 2 ** -7
    del yc, y0, d
    # (2) translation equivariance: a crop of the input gives the crop of the output, bit for bit (crop width a multiple
    # of 32, so that the crop takes the same -- channels-last -- kernels as the full image)
    crop = conv(Cplx(xr[:4, :, 5:105, 7:135].contiguous(), xi[:4, :, 5:105, 7:135].contiguous()))
    assert torch.equal(crop.real, y.real[:4, :, 5:103, 7:133]) and torch.equal(crop.imag, y.imag[:4, :, 5:103, 7:133])
    # (3) sampled outputs against the float64 oracle (bf16 output rounding)
    f = np.float64
    sub = (slice(250, 252), slice(None), slice(100, 110), slice(200, 212))
    rr, ri = orc.cplx_conv2d(N(xr[sub].float()).astype(f), N(xi[sub].float()).astype(f),
                             N(w.real.bfloat16().float()).astype(f), N(w.imag.bfloat16().float()).astype(f),
                             N(b.real).astype(f), N(b.imag).astype(f))
    got_r, got_i = N(y.real[250:252, :, 100:108, 200:210].float()), N(y.imag[250:252, :, 100:108, 200:210].float())
    np.testing.assert_allclose(got_r, rr, rtol=1e-2, atol=1e-2 * np.abs(rr).max())
    np.testing.assert_allclose(got_i, ri, rtol=1e-2, atol=1e-2 * np.abs(ri).max())
    # (4) batch-norm whitens: per channel, the output has zero mean and covariance W W^T (W = the
    # layer's 2x2 affine), and the running statistics moved by momentum * batch statistics
    bn = nn.CplxBatchNorm2d(C).to(DEV)
    z = bn(y)
    zr, zi = z.real.float(), z.imag.float()
    dims = (0, 2, 3)
    assert float(zr.mean(dims).abs().max()) < 2e-3 and float(zi.mean(dims).abs().max()) < 2e-3
    W = bn.weight.detach()                                       # [2, 2, C]
    want_uu = W[0, 0] ** 2 + W[0, 1] ** 2
    want_vv = W[1, 0] ** 2 + W[1, 1] ** 2
    want_uv = W[0, 0] * W[1, 0] + W[0, 1] * W[1, 1]
    np.testing.assert_allclose(N((zr * zr).mean(dims)), N(want_uu), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(N((zi * zi).mean(dims)), N(want_vv), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(N((zr * zi).mean(dims)), N(want_uv), rtol=2e-2, atol=2e-3)
    mean_r = y.real.float().mean(dims)
    np.testing.assert_allclose(N(bn.running_mean[0]), 0.1 * N(mean_r), rtol=1e-3, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1
    # (5) full-size backward is finite and the weight gradient scales exactly with the upstream one
    del z, zr, zi

    def wgrad(scale_):
        conv.zero_grad(set_to_none=True)
        out = conv(Cplx(xr, xi))
        torch.autograd.backward((out.real, out.imag), (y.real * scale_, y.imag * scale_))
        return conv.weight.real.grad.clone(), conv.bias.real.grad.clone()
    g1, b1 = wgrad(1.0)
    g2, b2 = wgrad(2.0)
    assert torch.isfinite(g1).all()
    np.testing.assert_allclose(N(g2), 2 * N(g1), rtol=1e-6)      # split-K slabs: float32 sums, exact x2
    np.testing.assert_allclose(N(b2), 2 * N(b1), rtol=1e-6)


# ---- round 4: the full-size BACKWARD legs, value-checked ------------------------------------------
def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().double().numpy()


def _cplx_noise_at(elems, seed, offset):
    """The in-kernel noise of a complex layer at linear output indices `elems` (oracle/philox.py: element e is pair e & 1
    of Philox group e >> 1, scaled by sqrt(1/2))."""
    from oracle import philox
    e = np.asarray(elems, dtype=np.uint64)
    x = philox.philox4x32(e >> np.uint64(1), offset, seed)
    u = philox._u01(x)
    odd = (e & np.uint64(1)).astype(bool)
    u0, u1 = np.where(odd, u[:, 2], u[:, 0]), np.where(odd, u[:, 3], u[:, 1])
    r = np.sqrt(-2 * np.log(u0)) * np.sqrt(0.5)
    return r * np.cos(2 * np.pi * u1), r * np.sin(2 * np.pi * u1)


def test_cfg4_lrt_backward_values_at_full_size():
    """configs[3] at its own batch 2^20: the input gradient (2^31 elements per plane -- the int32 boundary -- incl. the fused
    2 x g_a term), dW and dlog_sigma2 against float64 on sampled rows / columns, the LAST rows included.  The noise of the
    sampled outputs comes from the numpy statement of the Philox stream (reference: nn/relevance/complex/base.py:43-56
    differentiated)."""
    from cplxmodule_amd import Cplx
    from cplxmodule_amd.nn import relevance as rel
    B, F = 1 << 20, 2048
    torch.manual_seed(0)
    layer = rel.CplxLinearVD(F, F).to(DEV)
    with torch.no_grad():
        layer.log_sigma2.uniform_(-12, 4)
    xr, xi = _bf(B, F, seed=11).requires_grad_(True), _bf(B, F, seed=12).requires_grad_(True)
    rel.noise.manual_seed(9)
    y = layer(Cplx(xr, xi))                          # (seed 9, offset 1)
    gr, gi = y.real.detach().clone(), y.imag.detach().clone()     # upstream gradient of sum |y|^2 / 2
    torch.autograd.backward((y.real, y.imag), (gr, gi))
    dxr, dxi = xr.grad, xi.grad
    assert dxr.shape == (B, F) and torch.isfinite(dxr.float()).all() and torch.isfinite(dxi.float()).all()
    f = np.float64
    w = layer.weight
    Wr, Wi = N(w.real.bfloat16().float()).astype(f), N(w.imag.bfloat16().float()).astype(f)
    ls2 = N(layer.log_sigma2).astype(f)
    S16 = N(layer.log_sigma2.exp().bfloat16().float()).astype(f)                  # the variance GEMM's bf16 operand
    # ---- (1) dX on sampled rows, first / middle / the very last ones (row index >= 2^20 - 8)
    rows = np.array([0, 1, 4099, 524288, 777777] + list(range(B - 8, B)))
    tr = torch.from_numpy(rows).to(DEV)
    X_r, X_i = N(xr[tr].float()).astype(f), N(xi[tr].float()).astype(f)
    G_r, G_i = N(gr[tr].float()).astype(f), N(gi[tr].float()).astype(f)
    a = _bf16_round(X_r * X_r + X_i * X_i)
    s2 = _bf16_round(a @ S16.T)
    e_r, e_i = _cplx_noise_at((rows[:, None].astype(np.uint64) * np.uint64(F) + np.arange(F, dtype=np.uint64)[None]).ravel(), 9, 1)
    e_r, e_i = e_r.reshape(len(rows), F), e_i.reshape(len(rows), F)
    sd = np.sqrt(np.maximum(s2, 1e-8))
    gs2 = _bf16_round(np.where(s2 >= 1e-8, (G_r * e_r + G_i * e_i) * 0.5 / sd, 0.0))
    g_a = _bf16_round(gs2 @ S16)
    ref_r = _bf16_round(G_r @ Wr + G_i @ Wi) + 2 * X_r * g_a
    ref_i = _bf16_round(-G_r @ Wi + G_i @ Wr) + 2 * X_i * g_a
    sc = np.abs(ref_r).max()
    assert np.abs(2 * X_r * g_a).max() > 0.05 * sc            # the fused term is a visible part of what is checked
    np.testing.assert_allclose(N(dxr[tr].float()), ref_r, rtol=2e-2, atol=2e-2 * sc)
    np.testing.assert_allclose(N(dxi[tr].float()), ref_i, rtol=2e-2, atol=2e-2 * sc)
    # ---- (2) dW = G^T conj(X) on three output rows, float64 over the full batch (chunked on the device)
    o_rows = torch.tensor([0, 777, F - 1], device=DEV)
    dWr = torch.zeros(3, F, dtype=torch.float64, device=DEV)
    dWi = torch.zeros_like(dWr)
    CH = 1 << 16
    for c0 in range(0, B, CH):
        sl = slice(c0, c0 + CH)
        Gr, Gi = gr[sl][:, o_rows].double().t(), gi[sl][:, o_rows].double().t()
        Xr, Xi = xr[sl].detach().double(), xi[sl].detach().double()
        dWr += Gr @ Xr + Gi @ Xi
        dWi += Gi @ Xr - Gr @ Xi
    sw = float(dWr.abs().max())
    np.testing.assert_allclose(N(layer.weight.real.grad[o_rows]), dWr.cpu().numpy(), rtol=1e-3, atol=1e-4 * sw)
    np.testing.assert_allclose(N(layer.weight.imag.grad[o_rows]), dWi.cpu().numpy(), rtol=1e-3, atol=1e-4 * sw)
    # ---- (3) dlog_sigma2 = (gs2^T |x|^2) exp(log_sigma2) on the same three rows: gs2 of those COLUMNS for every batch row
    oc = o_rows.cpu().numpy()
    elems = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(F) + oc[None].astype(np.uint64)).ravel()
    er, ei = _cplx_noise_at(elems, 9, 1)
    er, ei = torch.from_numpy(er.reshape(B, 3)).to(DEV), torch.from_numpy(ei.reshape(B, 3)).to(DEV)
    S3 = layer.log_sigma2.detach().exp().bfloat16().double()[o_rows]               # [3, F]
    acc = torch.zeros(3, F, dtype=torch.float64, device=DEV)
    for c0 in range(0, B, CH):
        sl = slice(c0, c0 + CH)
        a_c = (xr[sl].detach().float() ** 2 + xi[sl].detach().float() ** 2).bfloat16().double()    # [CH, F]
        s2_c = (a_c @ S3.t()).float().bfloat16().double()                                          # [CH, 3]
        sd_c = s2_c.clamp_min(1e-8).sqrt()
        gsd = gr[sl][:, o_rows].double() * er[sl] + gi[sl][:, o_rows].double() * ei[sl]
        gs2_c = torch.where(s2_c >= 1e-8, gsd * 0.5 / sd_c, torch.zeros_like(gsd)).float().bfloat16().double()
        acc += gs2_c.t() @ a_c
    ref = (acc * layer.log_sigma2.detach().double()[o_rows].exp()).cpu().numpy()
    got = N(layer.log_sigma2.grad[o_rows])
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=5e-3 * np.abs(ref).max())


def test_cfg3_conv_backward_values_at_full_size():
    """configs[2] at batch 256: the data gradient of the LAST image (highest addresses of the 2 x 2.1 GB gradient planes)
    against the float64 oracle, sampled weight-gradient entries and the bias gradient against float64 sums over the whole
    batch (reference: cplx.py:729-742 differentiated)."""
    from cplxmodule_amd import Cplx, nn
    B, C, H = 256, 64, 256
    xr, xi = _bf(B, C, H, H, seed=21).requires_grad_(True), _bf(B, C, H, H, seed=22).requires_grad_(True)
    torch.manual_seed(1)
    conv = nn.CplxConv2d(C, C, 3).to(DEV)
    y = conv(Cplx(xr, xi))
    gr, gi = _bf(B, C, H - 2, H - 2, seed=23), _bf(B, C, H - 2, H - 2, seed=24)
    torch.autograd.backward((y.real, y.imag), (gr, gi))
    dxr, dxi = xr.grad, xi.grad
    assert dxr.shape == xr.shape and torch.isfinite(dxr.float()).all()
    f = np.float64
    w = conv.weight
    wr, wi = N(w.real.bfloat16().float()).astype(f), N(w.imag.bfloat16().float()).astype(f)
    # (1) dX of the last image, a window of it (the oracle's im2col of a whole 64 x 256 x 256 image is 300 MB: crop the
    # gradient to the rows that reach the window and place the result)
    b = B - 1
    h0, h1, w0, w1 = 100, 116, 230, 256                      # input window, touches the right border
    g_sl = (slice(b, b + 1), slice(None), slice(h0 - 2, h1), slice(w0 - 2, H - 2))
    Gr, Gi = N(gr[g_sl].float()).astype(f), N(gi[g_sl].float()).astype(f)
    # dX[h, w] = sum_{kh, kw} G[h - kh, w - kw] conj(W[kh, kw]): a "full" correlation of the cropped gradient
    hh, ww = Gr.shape[2] + 2, Gr.shape[3] + 2
    zx = np.zeros((1, C, hh, ww))
    bw = orc.cplx_conv2d_bwd(Gr, Gi, zx, zx, wr, wi, has_bias=False)
    # rows / columns of that result whose every tap lies inside the crop: drop the first 2 rows / columns
    ref_r, ref_i = bw["dxr"][0, :, 2:h1 - h0 + 2, 2:], bw["dxi"][0, :, 2:h1 - h0 + 2, 2:]
    got_r, got_i = N(dxr[b, :, h0:h1, w0:w1].float()), N(dxi[b, :, h0:h1, w0:w1].float())
    sc = np.abs(ref_r).max()
    np.testing.assert_allclose(got_r, ref_r[:, :, :w1 - w0], rtol=1e-2, atol=1e-2 * sc)
    np.testing.assert_allclose(got_i, ref_i[:, :, :w1 - w0], rtol=1e-2, atol=1e-2 * sc)
    # (2) sampled weight-gradient entries: dW[co, ci, kh, kw] = sum_{b, h, w} G[b, co, h, w] conj(X[b, ci, h + kh, w + kw])
    Ho = H - 2
    for co, ci, kh, kw in [(0, 0, 0, 0), (63, 63, 2, 2), (17, 40, 1, 2), (5, 31, 2, 0)]:
        Gr_, Gi_ = gr[:, co].double(), gi[:, co].double()
        Xr_, Xi_ = xr[:, ci, kh:kh + Ho, kw:kw + Ho].detach().double(), xi[:, ci, kh:kh + Ho, kw:kw + Ho].detach().double()
        want_r = float((Gr_ * Xr_ + Gi_ * Xi_).sum())
        want_i = float((Gi_ * Xr_ - Gr_ * Xi_).sum())
        norm = float((Gr_.abs() * Xr_.abs()).sum())           # size of the sum's terms: the float32 slab sums' error scale
        assert abs(float(conv.weight.real.grad[co, ci, kh, kw]) - want_r) <= 1e-6 * norm + 1e-3 * abs(want_r)
        assert abs(float(conv.weight.imag.grad[co, ci, kh, kw]) - want_i) <= 1e-6 * norm + 1e-3 * abs(want_i)
    # (3) bias gradient = sum of G over batch and pixels
    np.testing.assert_allclose(N(conv.bias.real.grad), gr.double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=2.0)
    np.testing.assert_allclose(N(conv.bias.imag.grad), gi.double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=2.0)
