"""3-d convolution and abs-max pooling composed from the 2-d kernels.

Reference: cplx.conv3d (cplxmodule/cplx.py:841-857), CplxConv3d (nn/modules/conv.py:199-247),
the 3-d VD / ARD / masked layers, cplx.max_pool3d (cplx.py:1193-1200).  Not on the hot path
(SURVEY 8 names conv2d), so there is no dedicated 3-d kernel: a [kd, kh, kw] correlation is the
sum over the kd depth taps of 2-d correlations of depth-strided slices, each of which runs the
conv2d kernels (forward, dgrad, wgrad through their autograd Functions) on a [B * D', C, H, W]
batch; abs-max pooling is separable (max modulus over (h, w), then over d) and runs the 2-d and
1-d pooling kernels.  The variance path of the VD layers is the same composition on
(|x|^2, exp(log_sigma2)) followed by the fused noise-injection kernel.
"""
import torch
import torch.nn.functional as F

from . import ops
from .cplx import Cplx
from .conv import CplxConv2dFn, RealConv2dFn


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def _depth_out(D, k, s, p, d):
    return (D + 2 * p - d * (k - 1) - 1) // s + 1


def _tap(x, a, dd, sd, Dout):
    """depth tap `a` of a [B, C, Dp, H, W] tensor as a 2-d batch [B * Dout, C, H, W]"""
    xs = x[:, :, a * dd: a * dd + sd * (Dout - 1) + 1: sd]
    B, C, _, H, W = xs.shape
    return xs.permute(0, 2, 1, 3, 4).reshape(B * Dout, C, H, W)


def _fold(y, B, Dout):
    """[B * Dout, O, H', W'] -> [B, O, Dout, H', W']"""
    return y.view(B, Dout, *y.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()


def _prepare(planes, kd, stride, padding, dilation, padding_mode):
    """pads what the 2-d kernels cannot (depth; everything for circular mode) -> planes, 2-d padding"""
    (sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = stride, padding, dilation
    if padding_mode == "circular":
        pads = []
        # symmetric_circular_padding (cplxmodule/cplx.py:699-712) hands `padding` to F.pad in its own
        # order and F.pad starts at the LAST dim: padding[0] wraps W, padding[2] wraps D
        for p in (pd, ph, pw):
            pads.extend(((p + 1) // 2, p // 2))
        planes = [F.pad(t, tuple(pads), mode="circular") for t in planes]
        pd = ph = pw = 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    elif pd:
        planes = [F.pad(t, (0, 0, 0, 0, pd, pd)) for t in planes]
    Dout = _depth_out(planes[0].shape[2], kd, sd, 0, dd)
    return planes, (ph, pw), Dout


def cplx_conv3d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                padding_mode="zeros"):
    stride, padding, dilation = _triple(stride), _triple(padding), _triple(dilation)
    kd = weight.shape[2]
    (xr, xi), pad2, Dout = _prepare([input.real, input.imag], kd, stride, padding, dilation, padding_mode)
    B = xr.shape[0]
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr = yi = None
    for a in range(kd):
        tr, ti = CplxConv2dFn.apply(_tap(xr, a, dilation[0], stride[0], Dout),
                                    _tap(xi, a, dilation[0], stride[0], Dout),
                                    weight.real[:, :, a], weight.imag[:, :, a],
                                    br if a == 0 else None, bi if a == 0 else None,
                                    stride[1:], pad2, dilation[1:], groups, False)
        yr, yi = (tr, ti) if yr is None else (yr + tr, yi + ti)
    return Cplx(_fold(yr, B, Dout), _fold(yi, B, Dout))


def real_conv3d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    stride, padding, dilation = _triple(stride), _triple(padding), _triple(dilation)
    kd = w.shape[2]
    (x,), pad2, Dout = _prepare([x], kd, stride, padding, dilation, "zeros")
    B = x.shape[0]
    y = None
    for a in range(kd):
        t = RealConv2dFn.apply(_tap(x, a, dilation[0], stride[0], Dout), w[:, :, a],
                               b if a == 0 else None, stride[1:], pad2, dilation[1:], groups)
        y = t if y is None else y + t
    return _fold(y, B, Dout)


def _conv_args(layer):
    return layer.stride, layer.padding, layer.dilation, layer.groups


def cplx_conv3d_lrt(layer, input, eps=None):
    """Training-mode forward of CplxConv3dVD / ARD (complex/base.py:120-135 with F.conv3d)."""
    mu = cplx_conv3d(input, layer.weight, layer.bias, *_conv_args(layer))
    a = ops.Abs2Fn.apply(input.real, input.imag)
    S = ops.ExpFn.apply(layer.log_sigma2).to(a.dtype)
    s2 = real_conv3d(a, S, None, *_conv_args(layer)).float()
    if eps is not None:
        er, ei, seed, offset = eps.real.contiguous(), eps.imag.contiguous(), 0, 0
    else:
        er, ei, seed, offset = layer._draw_noise(mu.shape, input)
    return Cplx(*ops.ReparamFn.apply(mu.real, mu.imag, s2, er, ei, seed, offset))


def real_conv3d_layer(layer, input, eps=None):
    """Forward of Conv3dVD / ARD (real/base.py:116-163): eval -> mean only."""
    mu = real_conv3d(input, layer.weight, layer.bias, *_conv_args(layer))
    if not layer.training:
        return mu
    a = ops.Abs2Fn.apply(input, None)
    S = ops.ExpFn.apply(layer.log_sigma2).to(a.dtype)
    s2 = real_conv3d(a, S, None, *_conv_args(layer)).float()
    seed = offset = 0
    if eps is None:
        eps, seed, offset = layer._draw_noise(mu.shape, input)
    return ops.ReparamFn.apply(mu, None, s2, eps, None, seed, offset)[0]


def cplx_max_pool3d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """[B, C, D, H, W]: largest modulus over (h, w) per depth slice, then over the depth window --
    the same element (first maximum in (d, h, w) order) a direct 3-d scan selects."""
    from . import cplx
    k = _triple(kernel_size)
    s = k if stride is None else _triple(stride)
    p, d = _triple(padding), _triple(dilation)
    B, C, D, H, W = input.shape
    flat = Cplx(input.real.reshape(B, C * D, H, W), input.imag.reshape(B, C * D, H, W))
    y = cplx.max_pool2d(flat, k[1:], s[1:], p[1:], d[1:], ceil_mode)
    Ho, Wo = y.shape[-2:]

    def depth_last(t):
        return t.view(B, C, D, Ho, Wo).permute(0, 1, 3, 4, 2).reshape(B, C * Ho * Wo, D)

    z = cplx.max_pool1d(Cplx(depth_last(y.real), depth_last(y.imag)), k[0], s[0], p[0], d[0], ceil_mode)
    Do = z.shape[-1]

    def depth_back(t):
        return t.view(B, C, Ho, Wo, Do).permute(0, 1, 4, 2, 3).contiguous()

    return Cplx(depth_back(z.real), depth_back(z.imag))
