"""Complex / real 2-d convolution on the implicit-GEMM kernels (csrc/conv.hip), with autograd,
and the conv flavours of the local-reparameterization layers.

Reference: cplx.conv2d -> convnd (cplxmodule/cplx.py:770-838), CplxConvNdGaussianMixin
(nn/relevance/complex/base.py:120-135), ConvNdGaussianMixin (nn/relevance/real/base.py:116-163).
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import call, try_call, dtype_code, ptr, require_device, stream_ptr
from .cplx import Cplx

_ws_cache = {}


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _geom(x_shape, w_shape, stride, padding, dilation, groups):
    B, Ci, H, W = x_shape
    Co, _, KH, KW = w_shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    arr = (ctypes.c_int * 14)(B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups)
    Ho = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    if Ho <= 0 or Wo <= 0:
        raise ValueError("convolution output would be empty")
    return arr, (B, Co, Ho, Wo)


def _scratch(device, nbytes):
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _ws_cache[key] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    return buf


_ktab_cache = {}


def _ktab(geom, mode, device):
    """Device copy of the (offset, dh, dw) table of the bf16 fast path (cached per geometry)."""
    key = (tuple(geom), mode, device.index)
    tab = _ktab_cache.get(key)
    if tab is None:
        lib = _lib.load()
        n = int(lib.cplxamd_conv2d_ktab_size(geom, mode))
        host = (ctypes.c_int * n)()
        call("cplxamd_conv2d_ktab_fill", geom, mode, host)
        tab = _ktab_cache[key] = torch.tensor(list(host), dtype=torch.int32, device=device)
    return tab


def _repack_dgrad(w, groups):
    """[Co, Ci/g, KH, KW] -> [g][Ci/g][Co/g * KH * KW] (K-contiguous rows for the dgrad GEMM)."""
    Co, Cg, KH, KW = w.shape
    return w.view(groups, Co // groups, Cg, KH, KW).permute(0, 2, 1, 3, 4).contiguous()


def nhwc_pad(x, ph, pw, Hp=None, Wp=None, tail_rows=0):
    """planar NCHW bf16 -> zero-padded channels-last [B, Hp, Wp, C] (default: symmetric padding).
    tail_rows: extra zero rows of C appended after the last image (the wgrad kernel's K padding)."""
    B, C, H, W = x.shape
    Hp = H + 2 * ph if Hp is None else Hp
    Wp = W + 2 * pw if Wp is None else Wp
    rows = B * Hp * Wp
    buf = torch.empty(rows + tail_rows, C, dtype=torch.bfloat16, device=x.device)
    call("cplxamd_nhwc_pad", ptr(x), ptr(buf), B, C, H, W, ph, pw, Hp, Wp, stream_ptr())
    if tail_rows:
        buf[rows:].zero_()
    return buf[:rows].view(B, Hp, Wp, C)


def _rows_ok(geom, C, ph, pw):
    """Shapes the shifted-row kernel (csrc/conv_nhwc.hip) takes: stride 1, groups 1, C % 32 == 0."""
    B, H, W = geom[0], geom[3], geom[4]
    return (geom[7] == 1 and geom[8] == 1 and geom[13] == 1 and C % 32 == 0 and ph >= 0 and pw >= 0
            and (geom[6] - 1) * geom[12] <= 32 and B * (H + 2 * ph) * (W + 2 * pw) < 2 ** 31)


def _pack_rows(w, swap):
    """[Co, Ci, KH, KW] -> [KH][KW][C/16][N][16] for the shifted-row kernel: (C, N) = (Ci, Co) for
    the forward, (Co, Ci) with both spatial dims flipped for the data gradient (`swap`)."""
    if swap:
        w = w.flip(2, 3).transpose(0, 1)
    N, C, KH, KW = w.shape
    return w.reshape(N, C // 16, 16, KH, KW).permute(3, 4, 1, 0, 2).contiguous()


def _conv_rows(xr, xi, wpr, wpi, br, bi, yr, yi, C, Cout, geom, ph, pw, conj):
    """One launch of the shifted-row kernel on (padded copies of) xr / xi."""
    xpr = nhwc_pad(xr, ph, pw)
    xpi = None if xi is None else nhwc_pad(xi, ph, pw)
    B, Hp, Wp = xpr.shape[0], xpr.shape[1], xpr.shape[2]
    return try_call("cplxamd_conv2d_nhwc", ptr(xpr), ptr(xpi), ptr(wpr), ptr(wpi), ptr(br), ptr(bi),
                    ptr(yr), ptr(yi), B, Hp, Wp, C, Cout, geom[5], geom[6], geom[11], geom[12],
                    int(conj), dtype_code(yr), stream_ptr())


def conv_fwd(xr, xi, wr, wi, br, bi, geom, out_shape):
    yr = torch.empty(out_shape, dtype=xr.dtype, device=xr.device)
    yi = None if xi is None else torch.empty_like(yr)
    if xr.dtype == torch.bfloat16 and _rows_ok(geom, geom[1], geom[9], geom[10]):
        wpr = _pack_rows(wr, False)
        wpi = None if wi is None else _pack_rows(wi, False)
        if _conv_rows(xr, xi, wpr, wpi, br, bi, yr, yi, geom[1], geom[2], geom, geom[9], geom[10],
                      False):
            return yr, yi
    if xr.dtype == torch.bfloat16 and try_call(
            "cplxamd_conv2d_bf16_fwd", ptr(xr), ptr(xi), ptr(wr), ptr(wi), ptr(br), ptr(bi), ptr(yr),
            ptr(yi), geom, ptr(_ktab(geom, 0, xr.device)), stream_ptr()):
        return yr, yi
    call("cplxamd_conv2d_fwd", ptr(xr), ptr(xi), ptr(wr), ptr(wi), ptr(br), ptr(bi), ptr(yr),
         ptr(yi), geom, dtype_code(xr), stream_ptr())
    return yr, yi


def conv_dgrad(gr, gi, wr, wi, geom, x_shape):
    dxr = torch.empty(x_shape, dtype=gr.dtype, device=gr.device)
    dxi = None if gi is None else torch.empty_like(dxr)
    # full correlation of the padded output gradient with the flipped, conjugated weight
    qh, qw = (geom[5] - 1) * geom[11] - geom[9], (geom[6] - 1) * geom[12] - geom[10]
    ggeom = list(geom)
    ggeom[3], ggeom[4] = gr.shape[2], gr.shape[3]
    if gr.dtype == torch.bfloat16 and _rows_ok(ggeom, geom[2], qh, qw):
        wdr = _pack_rows(wr, True)
        wdi = None if wi is None else _pack_rows(wi, True)
        if _conv_rows(gr, gi, wdr, wdi, None, None, dxr, dxi, geom[2], geom[1], geom, qh, qw, True):
            return dxr, dxi
    if gr.dtype == torch.bfloat16 and geom[7] == 1 and geom[8] == 1:
        wtr = _repack_dgrad(wr, geom[13])
        wti = None if wi is None else _repack_dgrad(wi, geom[13])
        if try_call("cplxamd_conv2d_bf16_dgrad", ptr(gr), ptr(gi), ptr(wtr), ptr(wti), ptr(dxr),
                    ptr(dxi), geom, ptr(_ktab(geom, 1, gr.device)), stream_ptr()):
            return dxr, dxi
    call("cplxamd_conv2d_dgrad", ptr(gr), ptr(gi), ptr(wr), ptr(wi), ptr(dxr), ptr(dxi), geom,
         dtype_code(gr), stream_ptr())
    return dxr, dxi


def _wgrad_rows(gr, gi, xr, xi, geom, w_shape, emul):
    """Weight gradient on channels-last copies (csrc/conv_nhwc_wgrad.hip); False if not eligible."""
    B, Ci, Co, KH, KW = geom[0], geom[1], geom[2], geom[5], geom[6]
    if not (_rows_ok(geom, 32, geom[9], geom[10]) and Ci % 8 == 0 and Co % 8 == 0 and KW <= 4):
        return None
    cplx = gi is not None
    Hp, Wp = geom[3] + 2 * geom[9], geom[4] + 2 * geom[10]
    xtail = 32 + (KH - 1) * geom[11] * Wp + (KW - 1) * geom[12]
    xpr = nhwc_pad(xr, geom[9], geom[10], tail_rows=xtail)
    xpi = nhwc_pad(xi, geom[9], geom[10], tail_rows=xtail) if cplx else None
    tail = (-(B * Hp * Wp)) % 32
    gpr = nhwc_pad(gr, 0, 0, Hp, Wp, tail_rows=tail)
    gpi = nhwc_pad(gi, 0, 0, Hp, Wp, tail_rows=tail) if cplx else None
    nbytes = int(_lib.load().cplxamd_conv2d_nhwc_wgrad_ws_bytes(B, Hp, Wp, Ci, Co, KH, KW, int(cplx)))
    ws = _scratch(gr.device, nbytes)
    dwr = torch.empty(w_shape, dtype=torch.float32, device=gr.device)
    dwi = torch.empty_like(dwr) if cplx else None
    if try_call("cplxamd_conv2d_nhwc_wgrad", ptr(gpr), ptr(gpi), ptr(xpr), ptr(xpi), ptr(emul),
                ptr(dwr), ptr(dwi), B, Hp, Wp, Ci, Co, KH, KW, geom[11], geom[12], ptr(ws),
                ws.numel(), stream_ptr()):
        return dwr, dwi
    return None


def conv_wgrad(gr, gi, xr, xi, geom, w_shape, emul=None):
    lib = _lib.load()
    cplx = gi is not None
    if gr.dtype == torch.bfloat16:
        out = _wgrad_rows(gr, gi, xr, xi, geom, w_shape, emul)
        if out is not None:
            return out
    if gr.dtype == torch.bfloat16:
        nbytes = int(lib.cplxamd_conv2d_bf16_wgrad_ws_bytes(geom, int(cplx)))
        ws = _scratch(gr.device, nbytes)
        dwr = torch.empty(w_shape, dtype=torch.float32, device=gr.device)
        dwi = torch.empty_like(dwr) if cplx else None
        if try_call("cplxamd_conv2d_bf16_wgrad", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(emul),
                    ptr(dwr), ptr(dwi), geom, ptr(_ktab(geom, 0, gr.device)), ptr(ws), ws.numel(),
                    stream_ptr()):
            return dwr, dwi
    nbytes = int(lib.cplxamd_conv2d_wgrad_ws_bytes(geom, int(cplx)))
    ws = _scratch(gr.device, nbytes)
    dwr = torch.empty(w_shape, dtype=torch.float32, device=gr.device)
    dwi = torch.empty_like(dwr) if cplx else None
    call("cplxamd_conv2d_wgrad", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(emul), ptr(dwr), ptr(dwi),
         geom, dtype_code(gr), ptr(ws), ws.numel(), stream_ptr())
    return dwr, dwi


def chansum(g):
    """sum over (batch, spatial) per channel of an NCHW tensor -> float32 [C]."""
    B, C = g.shape[0], g.shape[1]
    S = g.numel() // (B * C)
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = _scratch(g.device, 64 * C * 8)
    call("cplxamd_chansum", ptr(g), ptr(out), B, C, S, dtype_code(g), ptr(ws), stream_ptr())
    return out


class CplxConv2dFn(torch.autograd.Function):
    """Zero-padded complex conv (A.1 algebra with cross-correlation)."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, stride, padding, dilation, groups):
        require_device(xr, xi, wr, wi, br, bi)
        xr, xi = xr.contiguous(), xi.contiguous()
        wcr, wci = ops.cast(wr.contiguous(), xr.dtype), ops.cast(wi.contiguous(), xr.dtype)
        geom, oshape = _geom(xr.shape, wr.shape, stride, padding, dilation, groups)
        b = (None, None) if br is None else (br.contiguous(), bi.contiguous())
        yr, yi = conv_fwd(xr, xi, wcr, wci, b[0], b[1], geom, oshape)
        ctx.save_for_backward(xr, xi, wcr, wci)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, br is not None, wr.shape
        return yr, yi

    @staticmethod
    def backward(ctx, gr, gi):
        xr, xi, wcr, wci = ctx.saved_tensors
        gr, gi = gr.contiguous(), gi.contiguous()
        need = ctx.needs_input_grad
        dxr = dxi = dwr = dwi = dbr = dbi = None
        if need[0] or need[1]:
            dxr, dxi = conv_dgrad(gr, gi, wcr, wci, ctx.geom, xr.shape)
        if need[2] or need[3]:
            dwr, dwi = conv_wgrad(gr, gi, xr, xi, ctx.geom, ctx.wshape)
        if ctx.has_bias and (need[4] or need[5]):
            dbr, dbi = chansum(gr), chansum(gi)
        return dxr, dxi, dwr, dwi, dbr, dbi, None, None, None, None


class RealConv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups):
        require_device(x, w, b)
        x = x.contiguous()
        wc = ops.cast(w.contiguous(), x.dtype)
        geom, oshape = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        y, _ = conv_fwd(x, None, wc, None, None if b is None else b.contiguous(), None, geom, oshape)
        ctx.save_for_backward(x, wc)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, b is not None, w.shape
        return y

    @staticmethod
    def backward(ctx, g):
        x, wc = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        dx = dw = db = None
        if need[0]:
            dx, _ = conv_dgrad(g, None, wc, None, ctx.geom, x.shape)
        if need[1]:
            dw, _ = conv_wgrad(g, None, x, None, ctx.geom, ctx.wshape)
        if ctx.has_bias and need[2]:
            db = chansum(g)
        return dx, dw, db, None, None, None, None


def _circular_pad(t, padding):
    """symmetric_circular_padding (cplxmodule/cplx.py:701-714): ((p+1)//2, p//2) per spatial
    dim; F.pad's tuple starts at the LAST dim, the reference feeds `padding` in that order."""
    pads = []
    for p in _pair(padding):
        pads.extend(((p + 1) // 2, p // 2))
    return F.pad(t, tuple(pads), mode="circular")


def cplx_conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                padding_mode="zeros"):
    xr, xi = input.real, input.imag
    if padding_mode == "circular":
        xr, xi, padding = _circular_pad(xr, padding), _circular_pad(xi, padding), 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr, yi = CplxConv2dFn.apply(xr, xi, weight.real, weight.imag, br, bi, stride, padding,
                                dilation, groups)
    return Cplx(yr, yi)


class CplxConv2dLRTFn(torch.autograd.Function):
    """mu conv + variance conv + noise injection; backward per SURVEY A.2 with conv."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i, seed, offset, stride, padding,
                dilation, groups):
        require_device(xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i)
        xr, xi = xr.contiguous(), xi.contiguous()
        dt = xr.dtype
        wcr, wci = ops.cast(wr.contiguous(), dt), ops.cast(wi.contiguous(), dt)
        geom, oshape = _geom(xr.shape, wr.shape, stride, padding, dilation, groups)
        b = (None, None) if br is None else (br.contiguous(), bi.contiguous())
        mur, mui = conv_fwd(xr, xi, wcr, wci, b[0], b[1], geom, oshape)
        a = ops.abs2(xr, xi)
        S = ops.exp(ls2.contiguous(), out_dtype=dt)
        s2, _ = conv_fwd(a, None, S, None, None, None, geom, oshape)
        s2 = ops.cast(s2, torch.float32)
        eps = None if eps_r is None else (eps_r, eps_i)
        yr, yi = ops.reparam_fwd(mur, mui, s2, eps, seed, offset, inplace=True)
        ctx.save_for_backward(xr, xi, wcr, wci, ls2, s2, a, S, eps_r, eps_i)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, br is not None, wr.shape
        ctx.seed, ctx.offset = seed, offset
        return yr, yi

    @staticmethod
    def backward(ctx, gr, gi):
        xr, xi, wcr, wci, ls2, s2, a, S, eps_r, eps_i = ctx.saved_tensors
        gr, gi = gr.contiguous(), gi.contiguous()
        need = ctx.needs_input_grad
        eps = None if eps_r is None else (eps_r, eps_i)
        gs2 = ops.reparam_bwd(gr, gi, s2, eps, ctx.seed, ctx.offset, out_dtype=xr.dtype)
        dxr = dxi = dwr = dwi = dbr = dbi = dls2 = None
        if need[0] or need[1]:
            dxr, dxi = conv_dgrad(gr, gi, wcr, wci, ctx.geom, xr.shape)
            ga, _ = conv_dgrad(gs2, None, S, None, ctx.geom, xr.shape)
            ops.lrt_dx_accum(dxr, dxi, xr, xi, ga)
        if need[2] or need[3]:
            dwr, dwi = conv_wgrad(gr, gi, xr, xi, ctx.geom, ctx.wshape)
        if ctx.has_bias and (need[4] or need[5]):
            dbr, dbi = chansum(gr), chansum(gi)
        if need[6]:
            dls2, _ = conv_wgrad(gs2, None, a, None, ctx.geom, ctx.wshape,
                                 emul=ops.exp(ls2.contiguous()))
        return (dxr, dxi, dwr, dwi, dbr, dbi, dls2) + (None,) * 8


def cplx_conv2d_lrt(layer, input, eps=None):
    """Training-mode forward of CplxConv2dVD / ARD."""
    w, b = layer.weight, layer.bias
    br, bi = (None, None) if b is None else (b.real, b.imag)
    if eps is not None:
        er, ei, seed, offset = eps.real.contiguous(), eps.imag.contiguous(), 0, 0
    else:
        _, oshape = _geom(input.shape, w.shape, layer.stride, layer.padding, layer.dilation,
                          layer.groups)
        er, ei, seed, offset = layer._draw_noise(oshape, input)
    yr, yi = CplxConv2dLRTFn.apply(input.real, input.imag, w.real, w.imag, br, bi,
                                   layer.log_sigma2, er, ei, seed, offset, layer.stride,
                                   layer.padding, layer.dilation, layer.groups)
    return Cplx(yr, yi)


class RealConv2dLRTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, ls2, eps, seed, offset, stride, padding, dilation, groups):
        require_device(x, w, b, ls2, eps)
        x = x.contiguous()
        dt = x.dtype
        wc = ops.cast(w.contiguous(), dt)
        geom, oshape = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        mu, _ = conv_fwd(x, None, wc, None, None if b is None else b.contiguous(), None, geom, oshape)
        a = ops.abs2(x)
        S = ops.exp(ls2.contiguous(), out_dtype=dt)
        s2, _ = conv_fwd(a, None, S, None, None, None, geom, oshape)
        s2 = ops.cast(s2, torch.float32)
        y, _ = ops.reparam_fwd(mu, None, s2, eps, seed, offset, inplace=True)
        ctx.save_for_backward(x, wc, ls2, s2, a, S, eps)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, b is not None, w.shape
        ctx.seed, ctx.offset = seed, offset
        return y

    @staticmethod
    def backward(ctx, g):
        x, wc, ls2, s2, a, S, eps = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        gs2 = ops.reparam_bwd(g, None, s2, eps, ctx.seed, ctx.offset, out_dtype=x.dtype)
        dx = dw = db = dls2 = None
        if need[0]:
            dx, _ = conv_dgrad(g, None, wc, None, ctx.geom, x.shape)
            ga, _ = conv_dgrad(gs2, None, S, None, ctx.geom, x.shape)
            ops.lrt_dx_accum(dx, None, x, None, ga)
        if need[1]:
            dw, _ = conv_wgrad(g, None, x, None, ctx.geom, ctx.wshape)
        if ctx.has_bias and need[2]:
            db = chansum(g)
        if need[3]:
            dls2, _ = conv_wgrad(gs2, None, a, None, ctx.geom, ctx.wshape,
                                 emul=ops.exp(ls2.contiguous()))
        return (dx, dw, db, dls2) + (None,) * 7


def real_conv2d_layer(layer, input, eps=None):
    """Forward of Conv2dVD / ARD (eval: mean only; train: LRT)."""
    args = (layer.stride, layer.padding, layer.dilation, layer.groups)
    if not layer.training:
        return RealConv2dFn.apply(input, layer.weight, layer.bias, *args)
    seed = offset = 0
    if eps is None:
        _, oshape = _geom(input.shape, layer.weight.shape, *args)
        eps, seed, offset = layer._draw_noise(oshape, input)
    return RealConv2dLRTFn.apply(input, layer.weight, layer.bias, layer.log_sigma2, eps, seed,
                                 offset, *args)
