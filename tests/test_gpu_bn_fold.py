"""The batch-norm backward whose apply pass runs inside the weight-gradient launch of the convolution in front of it
(csrc/conv_cl_wgrad.hip FOLD, cplxamd_bn_bwd_coef + cplxamd_conv2d_cl_wgrad_bn_fl; bn.py / conv.py hand-over): against the
three separate launches, against the float64 oracle, and the cases in which the hand-over must NOT happen."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

dev = "cuda"


def _pair(B, Ci, Co, H, W, pad, seed=0, dilation=1):
    from cplxmodule_amd import Cplx, nn
    torch.manual_seed(seed)
    layer, bn = nn.CplxConv2d(Ci, Co, 3, padding=pad, dilation=dilation).to(dev), nn.CplxBatchNorm2d(Co).to(dev)
    with torch.no_grad():
        bn.weight.add_(0.3 * torch.randn_like(bn.weight)); bn.bias.add_(0.3 * torch.randn_like(bn.bias))
    mk = lambda: (torch.randn(B, Ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
                  .requires_grad_(True))
    return layer, bn, Cplx(mk(), mk())


def _grads(layer, bn, x, g):
    y = bn(layer(x))
    torch.autograd.backward((y.real, y.imag), g)
    out = [x.real.grad, x.imag.grad, layer.weight.real.grad, layer.weight.imag.grad, layer.bias.real.grad,
           layer.bias.imag.grad, bn.weight.grad, bn.bias.grad]
    x.real.grad = x.imag.grad = None
    layer.zero_grad(); bn.zero_grad()
    return [t.clone() for t in out]


@pytest.fixture
def force_cl():
    from cplxmodule_amd import conv as cv
    old = cv._CL_FORCE, cv._BN_FOLD
    cv._CL_FORCE = True
    yield cv
    cv._CL_FORCE, cv._BN_FOLD = old


@pytest.mark.parametrize("shape", [(2, 64, 64, 64, 64, 0), (3, 64, 64, 40, 72, 1), (2, 128, 64, 66, 50, 1),
                                   (2, 64, 128, 48, 64, 0), (1, 64, 64, 130, 97, 0), (3, 128, 128, 64, 32, 1)])
@pytest.mark.parametrize("train", [True, False])
def test_fold_matches_the_separate_launches(force_cl, shape, train):
    cv = force_cl
    calls = []
    real = cv.cl_wgrad_bn
    cv.cl_wgrad_bn = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        res = {}
        for fold in (False, True):
            cv._BN_FOLD = fold
            layer, bn, x = _pair(*shape)
            if not train:
                bn(layer(x)); bn.eval()
            torch.manual_seed(1)
            Ho = shape[3] + 2 * shape[5] - 2
            Wo = shape[4] + 2 * shape[5] - 2
            g = tuple(torch.randn(shape[0], shape[2], Ho, Wo, device=dev).bfloat16()
                      .contiguous(memory_format=torch.channels_last) for _ in range(2))
            res[fold] = _grads(layer, bn, x, g)
    finally:
        cv.cl_wgrad_bn = real
    assert len(calls) == 1                                   # the folded launch ran (once: only with _BN_FOLD)
    a, b = res[False], res[True]
    for i, (u, v) in enumerate(zip(a, b)):
        u, v = u.float(), v.float()
        scale = float(u.abs().max())
        if i in (4, 5):
            # the convolution's bias gradient = sum of dX: analytically E sum(g) - N k (zero in training mode); the separate
            # launches sum the bf16-rounded dX instead, i.e. that value plus rounding noise of ~sqrt(N) half-ulps
            n = shape[0] * Ho * Wo
            noise = 4e-3 * float(a[0].float().abs().max()) * np.sqrt(n) + 1e-6
            assert float((u - v).abs().max()) <= max(noise, 5e-3 * scale), i
        elif i < 2:
            # dX: the data gradient of dY values that differ by one bf16 rounding step in a few places
            assert float((u - v).abs().max()) <= 8e-3 * scale, i
        elif i < 4:
            assert float((u - v).abs().max()) <= 3e-4 * scale, i
        else:
            assert torch.equal(u, v), i                      # the layer's own parameter gradients: the same launches


def test_fold_c_abi_dy_and_dw_are_consistent(force_cl):
    """At the C ABI: dy of the folded launch against cplxamd_bn_bwd_sums (one bf16 step at most, in few places), and its dW
    bit-equal to cplxamd_conv2d_cl_wgrad_fl run on that dy (the MFMAs saw exactly the values that were stored)."""
    from cplxmodule_amd import _lib, bn as bnmod, conv as cv
    from cplxmodule_amd._lib import call, ptr, stream_ptr, launch_flags
    torch.manual_seed(0)
    B, Ci, Co, H, W, pad = 2, 64, 64, 72, 80, 1
    P = B * H * W
    cl = lambda *s: torch.randn(*s, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
    xr, xi = cl(B, Ci, H, W), cl(B, Ci, H, W)
    zr, zi, gr, gi = (cl(B, Co, H, W) for _ in range(4))
    zr = (zr.float() * 1.5 + 0.7).bfloat16()
    w = torch.tensor([[1.2, 0.1], [0.1, 0.8]], device=dev).reshape(2, 2, 1).repeat(1, 1, Co).contiguous()
    saved = torch.empty(8, Co, device=dev)
    ws = bnmod._ws(torch.device(dev, 0), Co)
    yr, yi = torch.empty_like(zr), torch.empty_like(zi)
    rm, rv = torch.zeros(2, Co, device=dev), torch.ones(2, 2, Co, device=dev)
    b = torch.zeros(2, Co, device=dev)
    call("cplxamd_bn_fwd_ex", ptr(zr), ptr(zi), ptr(yr), ptr(yi), P, Co, 1, ptr(w), ptr(b), ptr(rm), ptr(rv), ptr(saved), 1,
         _lib.BF16, 0.1, 1e-5, None, ptr(ws), ws.numel(), stream_ptr())
    dw1, db1 = torch.empty(2, 2, Co, device=dev), torch.empty(2, Co, device=dev)
    dxr, dxi, s1 = torch.empty_like(zr), torch.empty_like(zi), torch.empty(2, Co, device=dev)
    call("cplxamd_bn_bwd_sums", ptr(gr), ptr(gi), ptr(zr), ptr(zi), ptr(dxr), ptr(dxi), P, Co, 1, ptr(w), ptr(saved), ptr(dw1),
         ptr(db1), 1, _lib.BF16, ptr(s1), ptr(ws), ws.numel(), stream_ptr())
    dw2, db2 = torch.empty_like(dw1), torch.empty_like(db1)
    coef, s2 = torch.empty(Co, 12, device=dev), torch.empty(2, Co, device=dev)
    call("cplxamd_bn_bwd_coef", ptr(gr), ptr(gi), ptr(zr), ptr(zi), P, Co, 1, ptr(w), ptr(saved), ptr(dw2), ptr(db2), 1,
         _lib.BF16, ptr(coef), ptr(s2), ptr(ws), ws.numel(), stream_ptr())
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    wws = torch.empty(int(_lib.load().cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)), dtype=torch.uint8, device=dev)
    dyr, dyi = torch.empty_like(zr), torch.empty_like(zi)
    dwr, dwi = torch.empty(Co, Ci, 3, 3, device=dev), torch.empty(Co, Ci, 3, 3, device=dev)
    call("cplxamd_conv2d_cl_wgrad_bn_fl", ptr(gr), ptr(gi), ptr(zr), ptr(zi), ptr(coef), ptr(xr), ptr(xi), ptr(dyr), ptr(dyi),
         ptr(dwr), ptr(dwi), B, H, W, Ci, Co, 3, 3, 1, 1, pad, pad, ptr(wws), wws.numel(), launch_flags(), stream_ptr())
    for u, v in ((dxr, dyr), (dxi, dyi)):
        d = (u.float() - v.float()).abs()
        ulp = 2.0 ** -7 * u.float().abs().clamp_min(1e-30)          # one bf16 step at the magnitude of the value
        assert bool((d <= ulp * 1.01 + 1e-30).all())
        assert float((d > 0).float().mean()) < 0.02                  # ... and in few places
    er, ei = torch.empty_like(dwr), torch.empty_like(dwi)
    call("cplxamd_conv2d_cl_wgrad_fl", ptr(dyr), ptr(dyi), ptr(xr), ptr(xi), None, ptr(er), ptr(ei), B, H, W, Ci, Co, 3, 3, 1, 1,
         pad, pad, ptr(wws), wws.numel(), launch_flags(), stream_ptr())
    assert torch.equal(er, dwr) and torch.equal(ei, dwi)
    # the analytic column sums of dX against the sums of the stored values: rounding noise apart
    assert float((s1 - s2).abs().max()) <= 4e-3 * float(dxr.float().abs().max()) * np.sqrt(P)
    del cv


def test_fold_against_the_float64_oracle(force_cl):
    from oracle import cplx_oracle as orc
    cv = force_cl
    cv._BN_FOLD = True
    shape = (2, 64, 64, 48, 64, 1)
    layer, bn, x = _pair(*shape, seed=3)
    torch.manual_seed(4)
    g = tuple(torch.randn(2, 64, 48, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(2))
    got = _grads(layer, bn, x, g)
    f = lambda t: t.detach().double().cpu().numpy()  # noqa: E731
    xr, xi = f(x.real), f(x.imag)
    wr, wi = f(layer.weight.real.bfloat16()), f(layer.weight.imag.bfloat16())
    yr, yi = orc.cplx_conv2d(xr, xi, wr, wi, f(layer.bias.real), f(layer.bias.imag), stride=1, padding=1)
    yr, yi = f(torch.from_numpy(yr).bfloat16()), f(torch.from_numpy(yi).bfloat16())      # (the layer stores bf16)
    bw = orc.cplx_batch_norm_bwd(f(g[0]), f(g[1]), yr, yi, None, None, f(bn.weight), training=True, eps=bn.eps)
    cw = orc.cplx_conv2d_bwd(bw["dxr"], bw["dxi"], xr, xi, wr, wi, stride=1, padding=1)
    for name, t in (("dxr", got[0]), ("dxi", got[1]), ("dwr", got[2]), ("dwi", got[3])):
        ref = cw[name]
        np.testing.assert_allclose(f(t), ref, rtol=0, atol=2e-2 * np.abs(ref).max(), err_msg=name)


def test_no_hand_over_when_the_convolution_output_has_another_consumer(force_cl):
    """y = conv(x) feeds the batch-norm layer AND the loss directly: autograd sums two gradients for y, the planes the
    convolution's backward receives are not the ones the folded launch wrote, and its weight gradient must not be used."""
    from cplxmodule_amd import Cplx
    cv = force_cl
    res = {}
    for fold in (False, True):
        cv._BN_FOLD = fold
        layer, bn, x = _pair(2, 64, 64, 64, 64, 1, seed=5)
        y = layer(x)
        z = bn(y)
        torch.manual_seed(6)
        g = torch.randn_like(y.real)
        loss = (z.real.float() * g).sum() + (z.imag.float() * g).sum() + (y.real.float() * g).sum() * 0.5
        loss.backward()
        res[fold] = [layer.weight.real.grad.clone(), layer.weight.imag.grad.clone(), x.real.grad.clone()]
    for u, v in zip(res[False], res[True]):
        assert float((u.float() - v.float()).abs().max()) <= 8e-3 * float(u.float().abs().max())
    del Cplx


def test_fold_with_dilation_and_with_a_frozen_weight(force_cl):
    """Dilation 2 (the row kernel's geometry: the weight-gradient windows are 36 rows wide) takes the folded launch too; a
    convolution whose weight does not ask for a gradient has no weight-gradient launch to fold into and keeps the layer's
    own apply pass."""
    cv = force_cl
    res = {}
    for fold in (False, True):
        cv._BN_FOLD = fold
        layer, bn, x = _pair(2, 64, 64, 48, 64, 2, seed=7, dilation=2)
        torch.manual_seed(8)
        g = tuple(torch.randn(2, 64, 48, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(2))
        res[fold] = _grads(layer, bn, x, g)
    for i, (u, v) in enumerate(zip(res[False][:4], res[True][:4])):
        assert float((u.float() - v.float()).abs().max()) <= (8e-3 if i < 2 else 3e-4) * float(u.float().abs().max()), i
    cv._BN_FOLD = True
    calls = []
    real = cv.cl_wgrad_bn
    cv.cl_wgrad_bn = lambda *a, **k: (calls.append(real(*a, **k)), calls[-1])[1]
    try:
        layer, bn, x = _pair(2, 64, 64, 64, 64, 1, seed=9)
        layer.weight.real.requires_grad_(False); layer.weight.imag.requires_grad_(False)
        y = bn(layer(x))
        torch.autograd.backward((y.real, y.imag), (torch.ones_like(y.real), torch.ones_like(y.imag)))
    finally:
        cv.cl_wgrad_bn = real
    assert calls == [None] and x.real.grad is not None and layer.weight.real.grad is None
