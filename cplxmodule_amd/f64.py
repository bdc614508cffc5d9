"""float64 models (the reference's `.double()` layers: /root/reference/tests/test_modules.py:88-127).

A PARITY mode.  The contractions -- complex / real GEMM, complex / real convolution with both gradients -- and the
exponential integral run on this library's float64 kernels (csrc/f64.hip: everything torch would hand to a vendor
library, or to scipy on the host); the elementwise algebra around them (|x|^2, exp, the noise injection, log-alpha, the
penalty formulas, batch-norm's whitening) is spelled with torch's elementwise / reduction ops in the reference's own order,
under autograd -- so every float64 layer is differentiable to any order, as in the reference.  Nothing here is tuned: no
fusion, no hipGraph-captured noise, no data-parallel bucket views.  float64 tensors reach these functions through
`ops.Route` (the autograd Functions' `.apply` dispatches on the dtype of its first tensor argument).

Reference arithmetic: cplx.py:634-648 (linear), :167-174 (matmul), :717-838 (conv), nn/relevance/complex/base.py:27-56,
real/base.py:23-49 (LRT, log_alpha), real/vd.py:54-76, real/ard.py:10-39, complex/vd.py:15-99, complex/ard.py:9-39,
extensions/complex.py:18-163 (penalties), nn/modules/batchnorm.py:62-123, 189-278 (batch-norm).
"""
import ctypes
import math

import numpy as np
import torch

from ._lib import CplxAmdError, call, ptr, require_device, stream_ptr

F64 = torch.float64


def _chk(*ts):
    require_device(*ts)
    for t in ts:
        if t is not None and t.dtype != F64:
            raise CplxAmdError(f"float64 path: expected float64 tensors, got {t.dtype} (mixed precision is not offered)")


# ------------------------------------------------------------------------------------------ #
#  GEMM                                                                                      #
# ------------------------------------------------------------------------------------------ #
def _gemm(ar, ai, br, bi, conj_b=False, bias=None):
    """C[z, m, n] = sum_k A[z, m, k] op(B[z, n, k]) (+ bias[n]) for [.., M, K] / [.., N, K] operands given with ANY strides
    (2-d or one leading batch dimension); ai / bi None: real."""
    _chk(ar, ai, br, bi)
    batched = ar.dim() == 3
    if not batched:
        ar, br = ar.unsqueeze(0), br.unsqueeze(0)
        ai, bi = (None if ai is None else ai.unsqueeze(0)), (None if bi is None else bi.unsqueeze(0))
    Z, M, K = ar.shape
    N = br.shape[1]
    if ai is not None and (ai.stride() != ar.stride() or bi.stride() != br.stride()):
        ar, ai, br, bi = ar.contiguous(), ai.contiguous(), br.contiguous(), bi.contiguous()
    cr = torch.empty(Z, M, N, dtype=F64, device=ar.device)
    ci = None if ai is None else torch.empty_like(cr)
    b_r, b_i = (None, None) if bias is None else bias
    call("cplxamd_gemm_f64", ptr(ar), ptr(ai), ar.stride(1), ar.stride(2), ar.stride(0), ptr(br), ptr(bi), br.stride(1),
         br.stride(2), br.stride(0), ptr(b_r), ptr(b_i), ptr(cr), ptr(ci), N, M * N, Z, M, N, K, int(conj_b), stream_ptr())
    if not batched:
        cr, ci = cr[0], (None if ci is None else ci[0])
    return cr, ci


def _T(t):
    return t.transpose(-1, -2)


class CGemmFn(torch.autograd.Function):
    """C = A B^T on planar complex operands, no conjugation; gradients dA = G conj(B), dB = G^T conj(A) through this
    Function itself (any order of differentiation runs the same kernel)."""

    @staticmethod
    def forward(ctx, ar, ai, br, bi):
        ctx.save_for_backward(ar, ai, br, bi)
        return _gemm(ar.detach(), ai.detach(), br.detach(), bi.detach())

    @staticmethod
    def backward(ctx, gr, gi):
        ar, ai, br, bi = ctx.saved_tensors
        if gr is None:
            gr = torch.zeros(*ar.shape[:-1], br.shape[-2], dtype=ar.dtype, device=ar.device)
        gi = torch.zeros_like(gr) if gi is None else gi
        dar = dai = dbr = dbi = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dar, dai = CGemmFn.apply(gr, gi, _T(br), -_T(bi))        # sum_n G[m, n] conj(B[n, k])
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dbr, dbi = CGemmFn.apply(_T(gr), _T(gi), _T(ar), -_T(ai))  # sum_m G[m, n] conj(A[m, k])
        return dar, dai, dbr, dbi


class RGemmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _gemm(a.detach(), None, b.detach(), None)[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = RGemmFn.apply(g, _T(b)) if ctx.needs_input_grad[0] else None
        db = RGemmFn.apply(_T(g), _T(a)) if ctx.needs_input_grad[1] else None
        return da, db


def cplx_linear(xr, xi, wr, wi, br, bi, algo=0, mask=None):
    """ops.CplxLinearFn for float64 (`mask`: the masked layers' weight * mask)."""
    _chk(xr, xi, wr, wi, br, bi, mask)
    if mask is not None:
        wr, wi = wr * mask, wi * mask
    I, O = wr.shape[1], wr.shape[0]
    yr, yi = CGemmFn.apply(xr.reshape(-1, I), xi.reshape(-1, I), wr, wi)
    if br is not None:
        yr, yi = yr + br, yi + bi
    return yr.reshape(*xr.shape[:-1], O), yi.reshape(*xr.shape[:-1], O)


def real_linear(x, w, b, mask=None):
    _chk(x, w, b, mask)
    if mask is not None:
        w = w * mask
    y = RGemmFn.apply(x.reshape(-1, w.shape[1]), w)
    if b is not None:
        y = y + b
    return y.reshape(*x.shape[:-1], w.shape[0])


def matmul2d(ar, ai, vr, vi):
    """[M, K] @ [K, N]."""
    return CGemmFn.apply(ar, ai, _T(vr), _T(vi))


def matmul_batched(ar, ai, vr, vi):
    """[Z, M, K] @ [Z, K, N]."""
    return CGemmFn.apply(ar, ai, _T(vr), _T(vi))


# ------------------------------------------------------------------------------------------ #
#  Ei, log-alpha, penalties, masks                                                           #
# ------------------------------------------------------------------------------------------ #
class ExpiFn(torch.autograd.Function):
    """torch_expi (complex/vd.py:15-44): Ei on the device in float64, backward g e^x / x (differentiable torch ops)."""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        xc = x.detach().contiguous()
        y = torch.empty_like(xc)
        call("cplxamd_expi_f64", ptr(xc), ptr(y), xc.numel(), stream_ptr())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x) / x


def expi(x):
    return ExpiFn.apply(x)


def mask_mul(wr, wi, mask):
    """ops.MaskMulFn for float64."""
    return (wr * mask, wi * mask) if wi is not None else wr * mask


def cplx_abs(zr, zi):
    """Cplx.__abs__ (cplx.py:183-192): stack + norm, so the subgradient at 0 is torch's."""
    return torch.norm(torch.stack([zr, zi], dim=0), p=2, dim=0)


def log_alpha(ls2, wr, wi):
    """complex/base.py:27-31, real/base.py:23-26."""
    theta = torch.abs(wr) if wi is None else cplx_abs(wr, wi)
    return ls2 - 2 * torch.log(theta + 1e-12)


def penalty(kind, ls2, wr, wi):
    """Elementwise KL penalty of every kind (the files listed in the module docstring)."""
    _chk(ls2, wr, wi)
    sp, sg = torch.nn.functional.softplus, torch.sigmoid
    t = -log_alpha(ls2, wr, wi)
    if kind == "real_vd":
        return 0.5 * sp(t) + 0.63576 * sg(1.48695 * t - 1.87320)
    if kind == "real_ard":
        return 0.5 * sp(t)
    if kind == "cplx_ard":
        return sp(t)
    if kind == "cplx_vd":
        return np.euler_gamma + t - ExpiFn.apply(-torch.exp(t))
    if kind == "cplx_vd_approx":
        return sp(t) + 0.57810 * sg(1.36526 * t - 1.45926)
    if kind == "cplx_vd_scalefree":        # log|w| - ls2 - Ei(-1 / alpha) / 2 = t / 2 - ls2 / 2 - Ei(-e^t) / 2
        return 0.5 * t - 0.5 * ls2 - 0.5 * ExpiFn.apply(-torch.exp(t))
    if kind == "cplx_vd_bogus":            # the value drops the Ei term, the slope is the exact one (extensions/complex.py:142-160)
        v = np.euler_gamma + t - ExpiFn.apply(-torch.exp(t))
        return t + (v - v.detach()) - (t - t.detach())
    raise CplxAmdError(f"unknown penalty kind {kind!r}")


def penalty_sum(kind, ls2, wr, wi):
    return penalty(kind, ls2, wr, wi).sum()


def relevance_mask(wr, wi, ls2, threshold, count=False):
    with torch.no_grad():
        mask = (log_alpha(ls2, wr, wi) <= threshold).to(ls2.dtype)
        return (mask, mask.sum().to(torch.int64)) if count else mask


# ------------------------------------------------------------------------------------------ #
#  local reparameterization                                                                  #
# ------------------------------------------------------------------------------------------ #
def _noise(shape, like, complex_):
    """cplx.randn_like (cplx.py:544-562): ONE torch.randn(2, *shape) / sqrt 2; real layers randn(*shape)."""
    if complex_:
        e = torch.randn(2, *shape, dtype=like.dtype, device=like.device) / math.sqrt(2)
        return e[0], e[1]
    return torch.randn(*shape, dtype=like.dtype, device=like.device), None


def cplx_linear_lrt(xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i, seed=0, offset=0, kl_kind=None):
    """ops.CplxLinearLRTFn for float64 (complex/base.py:43-56); without a given noise tensor the draw is torch's."""
    mur, mui = cplx_linear(xr, xi, wr, wi, br, bi)
    s2 = real_linear(xr * xr + xi * xi, torch.exp(ls2), None)
    if eps_r is None:
        eps_r, eps_i = _noise(mur.shape, mur, True)
    sd = torch.sqrt(torch.clamp(s2, 1e-8))
    kl = penalty_sum(kl_kind, ls2, wr, wi) if kl_kind is not None else None
    return mur + eps_r.reshape(mur.shape) * sd, mui + eps_i.reshape(mui.shape) * sd, kl


def real_linear_lrt(x, w, b, ls2, eps, seed=0, offset=0, kl_kind=None):
    mu = real_linear(x, w, b)
    s2 = real_linear(x * x, torch.exp(ls2), None)
    if eps is None:
        eps, _ = _noise(mu.shape, mu, False)
    kl = penalty_sum(kl_kind, ls2, w, None) if kl_kind is not None else None
    return mu + eps.reshape(mu.shape) * torch.sqrt(torch.clamp(s2, 1e-8)), kl


# ------------------------------------------------------------------------------------------ #
#  convolution                                                                               #
# ------------------------------------------------------------------------------------------ #
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _geom(x_shape, w_shape, stride, padding, dilation, groups):
    B, Ci, H, W = x_shape
    Co, _, KH, KW = w_shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ho = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    if Ho <= 0 or Wo <= 0:
        raise ValueError("convolution output would be empty")
    return (ctypes.c_int * 14)(B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups), (B, Co, Ho, Wo)


def _conv(mode, pr, pi, qr, qi, bias, geom, out_shape):
    pr, qr = pr.contiguous(), qr.contiguous()
    pi, qi = (None if pi is None else pi.contiguous()), (None if qi is None else qi.contiguous())
    _chk(pr, pi, qr, qi)
    o_r = torch.empty(out_shape, dtype=F64, device=pr.device)
    o_i = None if pi is None else torch.empty_like(o_r)
    b_r, b_i = (None, None) if bias is None else bias
    call("cplxamd_conv2d_f64", ptr(pr), ptr(pi), ptr(qr), ptr(qi), ptr(b_r), ptr(b_i), ptr(o_r), ptr(o_i), geom, mode,
         stream_ptr())
    return o_r, o_i


class CplxConvFn(torch.autograd.Function):
    """Complex 2-d cross-correlation without bias; the gradients are the library's float64 data- / weight-gradient
    kernels (first order: a second differentiation raises)."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, stride, padding, dilation, groups):
        geom, oshape = _geom(xr.shape, wr.shape, stride, padding, dilation, groups)
        ctx.save_for_backward(xr, xi, wr, wi)
        ctx.geom = geom
        return _conv(0, xr.detach(), xi.detach(), wr.detach(), wi.detach(), None, geom, oshape)

    @staticmethod
    def backward(ctx, gr, gi):
        xr, xi, wr, wi = ctx.saved_tensors
        with torch.no_grad():
            gr = torch.zeros_like(gi) if gr is None else gr
            gi = torch.zeros_like(gr) if gi is None else gi
            dx = _conv(1, gr, gi, wr, wi, None, ctx.geom, xr.shape) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else (None, None)
            dw = _conv(2, gr, gi, xr, xi, None, ctx.geom, wr.shape) if (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) else (None, None)
        return _guard(dx[0], dx[1], dw[0], dw[1], None, None, None, None)


class RealConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation, groups):
        geom, oshape = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        ctx.save_for_backward(x, w)
        ctx.geom = geom
        return _conv(0, x.detach(), None, w.detach(), None, None, geom, oshape)[0]

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        with torch.no_grad():
            dx = _conv(1, g, None, w, None, None, ctx.geom, x.shape)[0] if ctx.needs_input_grad[0] else None
            dw = _conv(2, g, None, x, None, None, ctx.geom, w.shape)[0] if ctx.needs_input_grad[1] else None
        return _guard(dx, dw, None, None, None, None)


def _guard(*outs):
    """The convolution gradients come from raw kernels: under create_graph route them through an error node (as
    ops.once_differentiable does for the float32 / bf16 Functions)."""
    if not torch.is_grad_enabled():
        return outs
    live = [i for i, v in enumerate(outs) if isinstance(v, torch.Tensor)]
    if not live:
        return outs
    err = torch._C._functions.DelayedError(
        b"trying to differentiate twice a function that was marked with @once_differentiable (cplxmodule_amd float64 "
        b"convolution: gradients are computed by raw kernels)", len(live))
    wrapped = err(*[outs[i].detach().requires_grad_(True) for i in live])
    wrapped = (wrapped,) if isinstance(wrapped, torch.Tensor) else wrapped
    res = list(outs)
    for i, w in zip(live, wrapped):
        res[i] = w
    return tuple(res)


def cplx_conv2d(xr, xi, wr, wi, br, bi, stride=1, padding=0, dilation=1, groups=1, moments=False):
    """conv.CplxConv2dFn for float64."""
    yr, yi = CplxConvFn.apply(xr, xi, wr, wi, stride, padding, dilation, groups)
    if br is not None:
        yr, yi = yr + br.reshape(1, -1, 1, 1), yi + bi.reshape(1, -1, 1, 1)
    return yr, yi


def real_conv2d(x, w, b, stride=1, padding=0, dilation=1, groups=1):
    y = RealConvFn.apply(x, w, stride, padding, dilation, groups)
    return y if b is None else y + b.reshape(1, -1, 1, 1)


def cplx_conv2d_lrt(xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i, seed, offset, stride, padding, dilation, groups):
    """conv.CplxConv2dLRTFn for float64 (complex/base.py:120-135)."""
    mur, mui = cplx_conv2d(xr, xi, wr, wi, br, bi, stride, padding, dilation, groups)
    s2 = real_conv2d(xr * xr + xi * xi, torch.exp(ls2), None, stride, padding, dilation, groups)
    if eps_r is None:
        eps_r, eps_i = _noise(mur.shape, mur, True)
    sd = torch.sqrt(torch.clamp(s2, 1e-8))
    return mur + eps_r.reshape(mur.shape) * sd, mui + eps_i.reshape(mui.shape) * sd


def real_conv2d_lrt(x, w, b, ls2, eps, seed, offset, stride, padding, dilation, groups):
    mu = real_conv2d(x, w, b, stride, padding, dilation, groups)
    s2 = real_conv2d(x * x, torch.exp(ls2), None, stride, padding, dilation, groups)
    if eps is None:
        eps, _ = _noise(mu.shape, mu, False)
    return mu + eps.reshape(mu.shape) * torch.sqrt(torch.clamp(s2, 1e-8))


# ------------------------------------------------------------------------------------------ #
#  batch-norm                                                                                #
# ------------------------------------------------------------------------------------------ #
def cplx_batch_norm(xr, xi, weight, bias, running_mean, running_var, training, momentum, eps, process_group=None,
                    tracked=None):
    """bn.CplxBatchNormFn for float64: whiten2x2 + affine in the reference's own order (batchnorm.py:62-123, 254-278);
    running statistics updated in place through .data, as there."""
    _chk(xr, xi, weight, bias, running_mean, running_var)
    if process_group not in (None, False):
        raise CplxAmdError("float64 batch-norm: synchronised statistics are not offered")
    if not training and running_mean is None:
        raise ValueError("evaluation mode requires running statistics")
    if tracked is not None:
        tracked += 1
    axes = (0,) + tuple(range(2, xr.dim()))
    shp = (1, xr.shape[1]) + (1,) * (xr.dim() - 2)
    if training or running_mean is None:
        mu, mv = xr.mean(axes), xi.mean(axes)
        if running_mean is not None:
            running_mean.data += momentum * (torch.stack([mu, mv]).data - running_mean.data)
    else:
        mu, mv = running_mean[0], running_mean[1]
    cu, cv = xr - mu.reshape(shp), xi - mv.reshape(shp)
    if training or running_var is None:
        vuu = (cu * cu).mean(axes) + eps
        vvv = (cv * cv).mean(axes) + eps
        vuv = (cu * cv).mean(axes)
        if running_var is not None:
            cov = torch.stack([vuu, vuv, vuv, vvv]).reshape(2, 2, -1)
            running_var.data += momentum * (cov.data - running_var.data)
    else:
        vuu, vuv, vvv = running_var[0, 0], running_var[0, 1], running_var[1, 1]
    s = torch.sqrt(vuu * vvv - vuv * vuv)
    t = s * torch.sqrt(vuu + 2 * s + vvv)
    p, q, w = (vvv + s) / t, -vuv / t, (vuu + s) / t
    zu = cu * p.reshape(shp) + cv * q.reshape(shp)
    zv = cu * q.reshape(shp) + cv * w.reshape(shp)
    if weight is not None:
        W = weight.reshape(2, 2, *shp)
        zu, zv = (zu * W[0, 0] + zv * W[0, 1] + bias[0].reshape(shp), zu * W[1, 0] + zv * W[1, 1] + bias[1].reshape(shp))
    return zu, zv

