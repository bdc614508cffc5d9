"""The persistent channels-last convolution kernel (csrc/conv_cl.hip: forward and data gradient on unpadded
[B H W][C] planes, borders by masked fragment addresses, out-of-tensor rows by the buffer range check) against the
float64 numpy oracle on the bf16-rounded operands, through the public operator (cplx.conv2d + autograd)."""
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
    "two_tiles": dict(B=2, Ci=32, Co=64, H=18, W=21, k=3, padding=1, dilation=1),
    "dil2_two_column_tiles": dict(B=1, Ci=64, Co=128, H=16, W=16, k=3, padding=2, dilation=2),
    "nine_tiles_18_stages": dict(B=3, Ci=96, Co=64, H=40, W=37, k=3, padding=1, dilation=1),
    "one_by_three": dict(B=2, Ci=96, Co=64, H=9, W=30, k=(1, 3), padding=(0, 1), dilation=1),
    "wgrad_cl_w32": dict(B=2, Ci=64, Co=64, H=8, W=32, k=3, padding=1, dilation=1),
    "wgrad_cl_dil2_tiles": dict(B=3, Ci=128, Co=64, H=5, W=64, k=3, padding=2, dilation=2),
    "wgrad_cl_one_row_images": dict(B=7, Ci=64, Co=128, H=1, W=96, k=3, padding=1, dilation=1),
    "valid_padding": dict(B=2, Ci=64, Co=64, H=9, W=32, k=3, padding=0, dilation=1),
    "half_padding_dil2": dict(B=2, Ci=64, Co=64, H=11, W=64, k=3, padding=(1, 0), dilation=(2, 1)),
    "valid_one_row_out": dict(B=3, Ci=32, Co=64, H=3, W=40, k=3, padding=0, dilation=1),
    "tiny_image": dict(B=5, Ci=32, Co=64, H=3, W=2, k=3, padding=1, dilation=1),
    "five_by_three_dil": dict(B=2, Ci=96, Co=64, H=14, W=15, k=(5, 3), padding=(4, 3), dilation=(2, 3)),
}


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
@pytest.mark.parametrize("case", list(CASES))
def test_cl_conv_vs_oracle(case, layout):
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import Cplx, cplx, conv
    cfg = CASES[case]
    rs = np.random.RandomState(len(case))
    B, Ci, Co, H, W = cfg["B"], cfg["Ci"], cfg["Co"], cfg["H"], cfg["W"]
    kh, kw = (cfg["k"], cfg["k"]) if isinstance(cfg["k"], int) else cfg["k"]
    xr, xi = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(B, Ci, H, W) + 0.2)
    wr, wi = bf16_round(rs.randn(Co, Ci, kh, kw) * 0.1), bf16_round(rs.randn(Co, Ci, kh, kw) * 0.1)
    br, bi = rs.randn(Co).astype(np.float32), rs.randn(Co).astype(np.float32)
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    txr, txi = q(xr), q(xi)
    if layout == "channels_last":
        txr, txi = txr.contiguous(memory_format=torch.channels_last), txi.contiguous(memory_format=torch.channels_last)
    txr, txi = txr.requires_grad_(True), txi.requires_grad_(True)
    twr, twi = T(wr).requires_grad_(True), T(wi).requires_grad_(True)
    tbr, tbi = T(br).requires_grad_(True), T(bi).requires_grad_(True)
    kw_ = dict(stride=1, padding=cfg["padding"], dilation=cfg["dilation"], groups=1)
    geom, _ = conv._geom(txr.shape, twr.shape, 1, cfg["padding"], cfg["dilation"], 1)
    assert conv._cl_ok(geom) and conv._cl_ok(geom, dgrad=True) == (Ci % 64 == 0 and (kh * (Co // 16)) % 6 == 0)
    if Ci % 64 == 0 and Co % 64 == 0 and (kh, kw) == (3, 3):
        assert conv._cl_wgrad_ok(geom)               # any image width
    y = cplx.conv2d(Cplx(txr, txi), Cplx(twr, twi), Cplx(tbr, tbi), **kw_)
    if not conv._cl_wgrad_ok(geom):                  # such a layer stays planar as a whole; the kernel itself is
        assert not y.real.is_contiguous(memory_format=torch.channels_last) or y.real.is_contiguous()   # checked below
        y0 = conv.cl_conv(txr.detach(), txi.detach(), q(wr), q(wi), tbr.detach(), tbi.detach(), geom)
        f = np.float64
        yr0, yi0 = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br.astype(f), bi.astype(f), **kw_)
        np.testing.assert_allclose(N(y0[0]), yr0, rtol=1e-2, atol=1e-2 * np.abs(yr0).max())
        np.testing.assert_allclose(N(y0[1]), yi0, rtol=1e-2, atol=1e-2 * np.abs(yi0).max())
        if conv._cl_ok(geom, dgrad=True):
            g0r, g0i = bf16_round(rs.randn(*yr0.shape)), bf16_round(rs.randn(*yr0.shape))
            d0 = conv.cl_conv(q(g0r), q(g0i), q(wr), q(wi), None, None, geom, dgrad=True)
            bw0 = orc.cplx_conv2d_bwd(g0r.astype(f), g0i.astype(f), xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), **kw_)
            np.testing.assert_allclose(N(d0[0]), bw0["dxr"], rtol=2e-2, atol=2e-2 * np.abs(bw0["dxr"]).max())
            np.testing.assert_allclose(N(d0[1]), bw0["dxi"], rtol=2e-2, atol=2e-2 * np.abs(bw0["dxi"]).max())
    else:
        assert y.real.is_contiguous(memory_format=torch.channels_last)      # the channels-last kernels ran
    f = np.float64
    yr, yi = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br.astype(f), bi.astype(f), **kw_)
    assert tuple(y.shape) == yr.shape
    np.testing.assert_allclose(N(y.real), yr, rtol=1e-2, atol=1e-2 * np.abs(yr).max())
    np.testing.assert_allclose(N(y.imag), yi, rtol=1e-2, atol=1e-2 * np.abs(yi).max())
    gr, gi = bf16_round(rs.randn(*yr.shape)), bf16_round(rs.randn(*yr.shape))
    ((y.real * q(gr)).sum() + (y.imag * q(gi)).sum()).backward()
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), **kw_)
    got = dict(dxr=txr.grad, dxi=txi.grad, dwr=twr.grad, dwi=twi.grad, dbr=tbr.grad, dbi=tbi.grad)
    for n, t in got.items():
        np.testing.assert_allclose(N(t), bw[n], rtol=2e-2, atol=2e-2 * np.abs(bw[n]).max(), err_msg=n)


def test_cl_conv_many_tiles_per_workgroup():
    """More tiles than CUs (the ring runs through tile boundaries, bias by LDS-DMA, dump rows): against aten's float32
    convolution of the same bf16 operands -- a checker only -- and exactly linear in the input."""
    from cplxmodule_amd import Cplx, cplx
    dev = "cuda"
    torch.manual_seed(5)
    B, C, H, W = 3, 64, 250, 256                # P = 192000 -> 377 tiles of 510 rows; 6000 wgrad stages
    xr, xi = torch.randn(B, C, H, W, device=dev).bfloat16(), torch.randn(B, C, H, W, device=dev).bfloat16()
    wr, wi = (torch.randn(64, C, 3, 3, device=dev) * 0.05).bfloat16().float(), (torch.randn(64, C, 3, 3, device=dev) * 0.05).bfloat16().float()
    br, bi = torch.randn(64, device=dev), torch.randn(64, device=dev)
    xr.requires_grad_(True); xi.requires_grad_(True); wr.requires_grad_(True); wi.requires_grad_(True)
    y = cplx.conv2d(Cplx(xr, xi), Cplx(wr, wi), Cplx(br, bi), padding=1)
    assert y.real.is_contiguous(memory_format=torch.channels_last)
    F = torch.nn.functional
    a, b = xr.detach().float(), xi.detach().float()
    wr_, wi_ = wr.detach(), wi.detach()
    ref_r = F.conv2d(a, wr_, padding=1) - F.conv2d(b, wi_, padding=1) + br.view(1, -1, 1, 1)
    ref_i = F.conv2d(a, wi_, padding=1) + F.conv2d(b, wr_, padding=1) + bi.view(1, -1, 1, 1)
    for got, ref in ((y.real, ref_r), (y.imag, ref_i)):
        err = (got.float() - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item(), err
    gr, gi = torch.randn_like(ref_r).bfloat16(), torch.randn_like(ref_r).bfloat16()
    torch.autograd.backward((y.real, y.imag), (gr, gi))
    g1, g2 = gr.float(), gi.float()
    # dX = conv_transpose(G, conj(W)): real part Gr*Wr + Gi*Wi, imaginary part Gi*Wr - Gr*Wi
    dref_r = F.conv_transpose2d(g1, wr_, padding=1) + F.conv_transpose2d(g2, wi_, padding=1)
    dref_i = F.conv_transpose2d(g2, wr_, padding=1) - F.conv_transpose2d(g1, wi_, padding=1)
    for got, ref in ((xr.grad, dref_r), (xi.grad, dref_i)):
        err = (got.float() - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item(), err
    # dW = G^H-correlation with X: real part cw(Gr, Xr) + cw(Gi, Xi), imaginary part cw(Gi, Xr) - cw(Gr, Xi)
    cw = lambda gg, xx: torch.nn.grad.conv2d_weight(xx, wr_.shape, gg, padding=1)  # noqa: E731
    wref_r = cw(g1, a) + cw(g2, b)
    wref_i = cw(g2, a) - cw(g1, b)
    for got, ref in ((wr.grad, wref_r), (wi.grad, wref_i)):
        err = (got.float() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item(), err


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 8, 32), (3, 40, 5, 24), (1, 136, 16, 16)])
def test_layout_round_trip_exact(B, C, H, W):
    """to_channels_last / from_channels_last are pure data movement (the LDS-tile transposes): bit-identical to aten's."""
    from cplxmodule_amd import conv
    torch.manual_seed(B + C)
    x = torch.randn(B, C, H, W, device="cuda").bfloat16()
    cl = conv.to_channels_last(x)
    assert cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(cl, x)
    assert torch.equal(cl.permute(0, 2, 3, 1).contiguous(), x.permute(0, 2, 3, 1).contiguous())
    back = conv.from_channels_last(cl)
    assert back.is_contiguous() and torch.equal(back, x)


def test_bias_gradient_from_batchnorm_backward_sums(monkeypatch):
    """conv -> batch-norm, both channels-last: the convolution's bias gradient is the column sum of the gradient the
    batch-norm backward writes; its apply pass leaves those sums on the gradient tensors (ops.attach_colsum) and the
    convolution picks them up instead of a separate pass.  Same numbers as the separate pass."""
    from cplxmodule_amd import Cplx, conv as cv, nn, ops
    # (the layer's own apply pass: with the apply folded into the weight gradient -- round 6, tests/test_gpu_bn_fold.py -- the
    #  sums come from the reduce pass analytically, i.e. without the rounding noise of the stored bf16 values)
    monkeypatch.setattr(cv, "_BN_FOLD", False)
    torch.manual_seed(3)
    dev = "cuda"
    conv_, bn = nn.CplxConv2d(64, 64, 3, padding=1).to(dev), nn.CplxBatchNorm2d(64).to(dev)
    mk = lambda: torch.randn(2, 64, 32, 64, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
    x = Cplx(mk().requires_grad_(True), mk().requires_grad_(True))
    g = (mk(), mk())
    used = []
    real_hint = ops.colsum_hint

    def run(hint):
        monkeypatch.setattr(ops, "colsum_hint", hint)
        conv_.zero_grad(set_to_none=True)
        y = bn(conv_(x))
        torch.autograd.backward((y.real, y.imag), g)
        return conv_.bias.real.grad.clone(), conv_.bias.imag.grad.clone()

    def spy(t):
        h = real_hint(t)
        used.append(h is not None)
        return h

    with_r, with_i = run(spy)
    assert used == [True, True]
    sep_r, sep_i = run(lambda t: None)
    for a, b in ((with_r, sep_r), (with_i, sep_i)):
        # (4096 terms of magnitude ~1 summed in float32 in two different orders)
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 5e-5


def test_activations_keep_channels_last():
    """modReLU (scalar, 1-element and per-channel thresholds, with gradients), abs and dropout are elementwise: on
    channels-last inputs they keep the layout (no NCHW round trip inside a conv -> bn -> activation -> conv chain) and
    give the same numbers as on plain contiguous copies of the same tensors."""
    from cplxmodule_amd import Cplx, cplx, ops
    torch.manual_seed(11)
    dev = "cuda"
    for dtype in (torch.float32, torch.bfloat16):
        a, b = torch.randn(2, 16, 6, 10, device=dev).to(dtype), torch.randn(2, 16, 6, 10, device=dev).to(dtype)
        g1, g2 = torch.randn_like(a), torch.randn_like(a)
        for tau in (0.3, torch.tensor(0.3, device=dev), (0.5 * torch.rand(1, 16, 1, 1, device=dev))):
            outs = []
            for fmt in (torch.contiguous_format, torch.channels_last):
                zr, zi = (t.contiguous(memory_format=fmt).clone().requires_grad_(True) for t in (a, b))
                th = tau.clone().requires_grad_(True) if isinstance(tau, torch.Tensor) else tau
                y = cplx.modrelu(Cplx(zr, zi), th)
                if fmt == torch.channels_last:
                    assert y.real.is_contiguous(memory_format=fmt) and not y.real.is_contiguous()
                torch.autograd.backward((y.real, y.imag), (g1.contiguous(memory_format=fmt), g2.contiguous(memory_format=fmt)))
                outs.append([y.real, y.imag, zr.grad, zi.grad] + ([th.grad] if isinstance(th, torch.Tensor) else []))
                if fmt == torch.channels_last:
                    assert zr.grad.is_contiguous(memory_format=fmt)
            for p, q in zip(*outs):
                tol = 0 if p.shape == a.shape else 1e-5 * float(q.abs().max()) + 1e-6     # (dtau: a sum, order differs)
                assert float((p.detach().float() - q.detach().float()).abs().max()) <= tol
        zr, zi = (t.contiguous(memory_format=torch.channels_last).clone().requires_grad_(True) for t in (a, b))
        m = abs(Cplx(zr, zi))
        assert m.is_contiguous(memory_format=torch.channels_last) and torch.equal(m, ops.modulus(a, b))
        m.backward(g1.contiguous(memory_format=torch.channels_last))
        zr2, zi2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        abs(Cplx(zr2, zi2)).backward(g1)
        assert torch.equal(zr.grad, zr2.grad) and torch.equal(zi.grad, zi2.grad)


def test_three_block_chain_matches_planar_path():
    """[conv 3x3 -> batch-norm -> modReLU] x 3 on a channels-last input: every tensor between the layers stays
    channels-last, and outputs / all gradients agree with the same network run on the planar (NCHW) kernels."""
    from cplxmodule_amd import Cplx, nn, conv
    dev = "cuda"

    def build():
        torch.manual_seed(21)
        layers = []
        for _ in range(3):
            layers += [nn.CplxConv2d(64, 64, 3, padding=1), nn.CplxBatchNorm2d(64), nn.CplxModReLU(0.1)]
        return torch.nn.Sequential(*layers).to(dev)

    torch.manual_seed(22)
    a, b = torch.randn(2, 64, 16, 32, device=dev).bfloat16(), torch.randn(2, 64, 16, 32, device=dev).bfloat16()
    ga, gb = torch.randn(2, 64, 16, 32, device=dev).bfloat16(), torch.randn(2, 64, 16, 32, device=dev).bfloat16()
    res = []
    for cl in (True, False):
        old = conv._CL_ENABLED
        conv._CL_ENABLED = cl
        try:
            net = build()
            fmt = torch.channels_last if cl else torch.contiguous_format
            xr, xi = (t.contiguous(memory_format=fmt).clone().requires_grad_(True) for t in (a, b))
            seen = []
            hooks = [m.register_forward_hook(lambda m_, i, o: seen.append(o.real.is_contiguous(memory_format=torch.channels_last)
                                                                          and not o.real.is_contiguous())) for m in net]
            y = net(Cplx(xr, xi))
            for h in hooks:
                h.remove()
            assert all(seen) if cl else not any(seen)
            torch.autograd.backward((y.real, y.imag), (ga.contiguous(memory_format=fmt), gb.contiguous(memory_format=fmt)))
            out = [y.real, y.imag, xr.grad, xi.grad]
            for name, p in net.named_parameters():
                # (a convolution's bias in front of a batch-norm has an analytically ZERO gradient -- the normalisation
                #  removes the mean -- so what either path computes there is rounding noise of the bf16 dX: not compared)
                if not (name.endswith(("bias.real", "bias.imag")) and p.dim() == 1):
                    out.append(p.grad)
            res.append([t.detach().float() for t in out])
        finally:
            conv._CL_ENABLED = old
    for p, q in zip(*res):
        assert p.shape == q.shape
        assert float((p - q).abs().max()) <= 4e-2 * float(q.abs().max()) + 1e-3, (p.shape, float((p - q).abs().max()), float(q.abs().max()))


def test_cl_entry_points_refuse_what_they_are_not_built_for():
    """The channels-last entry points answer CPLXAMD_ESHAPE / EALIGN / EINVAL for unsupported calls (the host then
    falls back to the padded-grid or generic kernels) instead of computing something else."""
    from cplxmodule_amd import _lib
    from cplxmodule_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    dev, bf = "cuda", torch.bfloat16
    B, C, H, W = 1, 64, 8, 32
    x = torch.zeros(B, H, W, C, device=dev, dtype=bf)
    y = torch.zeros(B, H, W, 64, device=dev, dtype=bf)
    wp = torch.zeros(int(lib.cplxamd_conv2d_cl_pack_bytes(64, C, 3, 3)), dtype=torch.uint8, device=dev)
    ws = torch.zeros(int(lib.cplxamd_conv2d_cl_ws_bytes(64)), dtype=torch.uint8, device=dev)

    def conv(C_=C, N=64, KH=3, KW=3, pad=1, dil=1, xin=x, mode=0, ws_=ws):
        return lib.cplxamd_conv2d_cl(ptr(xin), ptr(xin), ptr(wp), None, None, ptr(y), ptr(y), B, H, W, C_, N, KH, KW, dil, dil,
                                     pad, pad, mode, ptr(ws_), ws_.numel(), stream_ptr())
    class E:                                             # include/cplxamd.h
        EINVAL, EALIGN, ESHAPE = -1, -2, -3
    assert conv() == 0
    assert conv(KW=5, pad=2) == E.ESHAPE                 # kernel width
    assert conv(C_=24) == E.ESHAPE                       # channels not a multiple of 16
    assert conv(N=32) == E.ESHAPE                        # output channels not a multiple of 64
    assert conv(C_=32) == 0                              # KH * C / 16 = 6 is fine ...
    assert conv(C_=16) == E.ESHAPE                       # ... 3 is not a multiple of 6
    assert conv(pad=2) == E.ESHAPE                       # more than `same`
    assert conv(mode=2) == E.EINVAL
    assert conv(xin=x.view(-1)[4:].view(-1)) == E.EALIGN  # 8-byte aligned plane
    assert conv(ws_=ws[:16]) == E.EINVAL
    g = torch.zeros(B, H, W, 64, device=dev, dtype=bf)
    dw = torch.zeros(64, 64, 3, 3, device=dev)
    wsw = torch.zeros(int(lib.cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, 64, 64)), dtype=torch.uint8, device=dev)

    def wgrad(W_=W, Ci=64, KH=3, dil=1, pad=1):
        return lib.cplxamd_conv2d_cl_wgrad(ptr(g), ptr(g), ptr(x), ptr(x), None, ptr(dw), ptr(dw), B, H, W_, Ci, 64, KH, 3, dil, dil,
                                           pad, pad, ptr(wsw), wsw.numel(), stream_ptr())
    assert wgrad() == 0
    assert wgrad(Ci=32) == E.ESHAPE and wgrad(KH=1) == E.ESHAPE and wgrad(pad=2) == E.ESHAPE
    assert lib.cplxamd_cl_to_nchw(ptr(x), ptr(y), 1, 12, 64, stream_ptr()) == E.ESHAPE
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_channels_last_matches_planar(dtype):
    """Abs-max pooling on channels-last planes: same outputs (bit for bit), same gradients, layout kept."""
    from cplxmodule_amd import Cplx, cplx
    torch.manual_seed(5)
    a, b = torch.randn(2, 24, 13, 17, device="cuda").to(dtype), torch.randn(2, 24, 13, 17, device="cuda").to(dtype)
    for kw in (dict(kernel_size=2), dict(kernel_size=3, stride=2, padding=1), dict(kernel_size=(2, 3), stride=(1, 2), ceil_mode=True)):
        outs = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            zr, zi = (t.contiguous(memory_format=fmt).clone().requires_grad_(True) for t in (a, b))
            y = cplx.max_pool2d(Cplx(zr, zi), **kw)
            if fmt == torch.channels_last:
                assert y.real.is_contiguous(memory_format=fmt) and not y.real.is_contiguous()
            g1, g2 = torch.ones_like(y.real) * 0.5, torch.ones_like(y.imag) * 2
            torch.autograd.backward((y.real, y.imag), (g1, g2))
            outs.append((y.real.detach(), y.imag.detach(), zr.grad, zi.grad))
        for p, q in zip(*outs):
            assert torch.equal(p.contiguous(), q.contiguous())


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_pointwise_conv_runs_as_linear_layer(layout):
    """1 x 1 convolution on channels-last bf16 images = the complex GEMM on [B H W, C] rows: outputs and all gradients
    against the oracle, output channels-last."""
    from gpu_util import T, N, bf16_round
    from cplxmodule_amd import Cplx, cplx
    rs = np.random.RandomState(31)
    B, Ci, Co, H, W = 2, 64, 96, 9, 20
    xr, xi = bf16_round(rs.randn(B, Ci, H, W)), bf16_round(rs.randn(B, Ci, H, W))
    wr, wi = bf16_round(rs.randn(Co, Ci, 1, 1) * 0.1), bf16_round(rs.randn(Co, Ci, 1, 1) * 0.1)
    br, bi = rs.randn(Co).astype(np.float32), rs.randn(Co).astype(np.float32)
    q = lambda a: T(a, torch.bfloat16)  # noqa: E731
    txr, txi = q(xr), q(xi)
    if layout == "channels_last":
        txr, txi = txr.contiguous(memory_format=torch.channels_last), txi.contiguous(memory_format=torch.channels_last)
    txr, txi = txr.requires_grad_(True), txi.requires_grad_(True)
    twr, twi, tbr, tbi = (T(a).requires_grad_(True) for a in (wr, wi, br, bi))
    y = cplx.conv2d(Cplx(txr, txi), Cplx(twr, twi), Cplx(tbr, tbi))
    assert y.real.is_contiguous(memory_format=torch.channels_last) and not y.real.is_contiguous()
    f = np.float64
    yr, yi = orc.cplx_conv2d(xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f), br.astype(f), bi.astype(f))
    np.testing.assert_allclose(N(y.real), yr, rtol=1e-2, atol=1e-2 * np.abs(yr).max())
    np.testing.assert_allclose(N(y.imag), yi, rtol=1e-2, atol=1e-2 * np.abs(yi).max())
    gr, gi = bf16_round(rs.randn(*yr.shape)), bf16_round(rs.randn(*yr.shape))
    ((y.real * q(gr)).sum() + (y.imag * q(gi)).sum()).backward()
    bw = orc.cplx_conv2d_bwd(gr.astype(f), gi.astype(f), xr.astype(f), xi.astype(f), wr.astype(f), wi.astype(f))
    got = dict(dxr=txr.grad, dxi=txi.grad, dwr=twr.grad, dwi=twi.grad, dbr=tbr.grad, dbi=tbi.grad)
    for n, t in got.items():
        assert tuple(t.shape) == bw[n].shape
        np.testing.assert_allclose(N(t), bw[n], rtol=2e-2, atol=2e-2 * np.abs(bw[n]).max(), err_msg=n)


@pytest.mark.parametrize("seed", range(16))
def test_cl_conv_random_geometries_vs_float32_kernels(seed):
    """Property test: random stride-1 3 x 3 geometries (image sizes incl. tiny ones and widths that are / are not
    multiples of 32, paddings from 0 to `same`, dilations 1-2 per axis, 32-128 channels, 1-3 column tiles) through the
    channels-last kernels vs the exact float32 planar kernels on the same bf16-rounded operands: forward, data
    gradient, weight gradient, bias gradient."""
    from cplxmodule_amd import Cplx, cplx, conv
    rs = np.random.RandomState(500 + seed)
    B = int(rs.randint(1, 4))
    Ci, Co = int(rs.choice([32, 64, 96, 128])), int(rs.choice([64, 128, 192]))
    dh, dw = int(rs.randint(1, 3)), int(rs.randint(1, 3))
    ph, pw = int(rs.randint(0, dh + 1)), int(rs.randint(0, dw + 1))
    H = int(rs.randint(2 * dh + 1 - 2 * ph + 1 if 2 * dh + 1 - 2 * ph > 0 else 1, 40))
    W = int(rs.choice([32, 64, 96])) if seed % 4 == 0 else int(rs.randint(max(2 * dw + 1 - 2 * pw + 1, 2), 70))
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16().to("cuda")  # noqa: E731
    xr, xi = mk(B, Ci, H, W), mk(B, Ci, H, W)
    wr, wi = (mk(Co, Ci, 3, 3).float() * 0.1).bfloat16().float(), (mk(Co, Ci, 3, 3).float() * 0.1).bfloat16().float()
    br, bi = mk(Co).float(), mk(Co).float()
    kw_ = dict(stride=1, padding=(ph, pw), dilation=(dh, dw))
    geom, _ = conv._geom(xr.shape, wr.shape, 1, (ph, pw), (dh, dw), 1)
    outs = []
    for dt in (torch.bfloat16, torch.float32):
        leaves = [t.to(dt).clone().requires_grad_(True) for t in (xr, xi)] + [t.clone().requires_grad_(True) for t in (wr, wi, br, bi)]
        y = cplx.conv2d(Cplx(leaves[0], leaves[1]), Cplx(leaves[2], leaves[3]), Cplx(leaves[4], leaves[5]), **kw_)
        if dt == torch.bfloat16 and conv._cl_wgrad_ok(geom) and conv._cl_ok(geom):
            assert y.real.is_contiguous(memory_format=torch.channels_last)
        gg = torch.Generator(device="cpu").manual_seed(1000 + seed)
        gr = torch.randn(y.real.shape, generator=gg).bfloat16().to("cuda")
        gi = torch.randn(y.real.shape, generator=gg).bfloat16().to("cuda")
        torch.autograd.backward((y.real, y.imag), (gr.to(dt), gi.to(dt)))
        outs.append([y.real.float(), y.imag.float()] + [t.grad.float() for t in leaves])
    names = ["yr", "yi", "dxr", "dxi", "dwr", "dwi", "dbr", "dbi"]
    for n, a, b in zip(names, *outs):
        scale = float(b.detach().abs().max()) + 1e-6
        tol = 2e-2 if n[0] == "y" or n[1] == "x" else 1e-3       # bf16-rounded outputs vs fp32-accumulated gradients
        assert float((a.detach() - b.detach()).abs().max()) <= tol * scale, (n, seed, B, Ci, Co, dh, dw, ph, pw, H, W)
    # the forward / data-gradient kernel directly, also where the layer as a whole stays planar (image width not a
    # multiple of 32, 32 input channels)
    ref = outs[1]
    wb = (wr.bfloat16(), wi.bfloat16())
    if conv._cl_ok(geom):
        y2 = conv.cl_conv(xr, xi, wb[0], wb[1], br, bi, geom)
        for got, want in zip(y2, ref[:2]):
            assert float((got.float() - want.detach()).abs().max()) <= 2e-2 * float(want.detach().abs().max()) + 1e-6, ("fwd", seed)
    if conv._cl_ok(geom, dgrad=True):
        gg = torch.Generator(device="cpu").manual_seed(1000 + seed)
        gr = torch.randn(ref[0].shape, generator=gg).bfloat16().to("cuda")
        gi = torch.randn(ref[0].shape, generator=gg).bfloat16().to("cuda")
        d2 = conv.cl_conv(gr, gi, wb[0], wb[1], None, None, geom, dgrad=True)
        for got, want in zip(d2, ref[2:4]):
            assert float((got.float() - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-6, ("dgrad", seed)


@pytest.mark.parametrize("B,Ci,Co,H,W,pad", [
    (1, 32, 64, 3, 3, (1, 1)),            # one partial tile
    (2, 64, 64, 16, 32, (1, 1)),          # exactly one full tile per image
    (2, 32, 128, 17, 33, (1, 1)),         # one row / one column past the tile: four tiles, three of them slivers
    (3, 96, 64, 50, 70, (0, 1)),          # valid rows, same columns (output smaller than the input)
    (2, 64, 192, 35, 45, (1, 0)),
    (2, 128, 64, 20, 40, (0, 0)),
    (5, 64, 64, 130, 200, (1, 1)),        # 5 * 9 * 7 = 315 tiles x 1 column tile: more tiles than workgroups
    (9, 32, 128, 100, 100, (1, 1)),       # 9 * 7 * 4 = 252 tiles x 2 column tiles: two tiles per workgroup
])
def test_patch_kernel_matches_row_kernel(B, Ci, Co, H, W, pad):
    """conv_cl2.hip (2-D patch per 16-channel slice) against conv_cl.hip (row shifts, itself checked against the
    oracle above) on the same operands, forward and data gradient: the two accumulate the 9 * Ci terms in a different
    order in fp32 and round once to bf16, so they agree to one bf16 step of the largest output."""
    from cplxmodule_amd import conv
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16().to("cuda")  # noqa: E731
    wr, wi = (mk(Co, Ci, 3, 3).float() * 0.1).bfloat16(), (mk(Co, Ci, 3, 3).float() * 0.1).bfloat16()
    br, bi = mk(Co).float(), mk(Co).float()
    geom, _ = conv._geom((B, Ci, H, W), wr.shape, 1, pad, 1, 1)
    Ho, Wo = H + 2 * pad[0] - 2, W + 2 * pad[1] - 2
    cl = lambda *s: mk(*s).contiguous(memory_format=torch.channels_last)  # noqa: E731
    jobs = [("fwd", (cl(B, Ci, H, W), cl(B, Ci, H, W), wr, wi, br, bi, geom), {})]
    if Co % 32 == 0 and Ci % 64 == 0:
        jobs.append(("dgrad", (cl(B, Co, Ho, Wo), cl(B, Co, Ho, Wo), wr, wi, None, None, geom), dict(dgrad=True)))
    old = conv._CL_PATCH
    try:
        for name, a, k in jobs:
            conv._CL_PATCH = True
            seen = []
            real_try = conv.try_call
            def spy(n, *aa):
                seen.append((n, real_try(n, *aa)))
                return seen[-1][1]
            conv.try_call = spy
            try:
                got = conv.cl_conv(*a, **k)
            finally:
                conv.try_call = real_try
            assert seen == [("cplxamd_conv2d_cl2_fl", True)], name          # the patch kernel took it
            conv._CL_PATCH = False
            want = conv.cl_conv(*a, **k)
            for p, q in zip(got, want):
                assert p.shape == q.shape and bool(torch.isfinite(p.float()).all())
                assert float((p.float() - q.float()).abs().max()) <= 2 ** -7 * float(q.float().abs().max()), name
    finally:
        conv._CL_PATCH = old
