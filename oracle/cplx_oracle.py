"""CPU oracle: a numpy restatement of the cplxmodule hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``cplxmodule_amd`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / the timed CPU baseline.

Every function restates one reference function (cited as file:line under
``/root/reference``) in plain numpy.  Forward passes follow the reference's op
order; backward passes are the hand-derived formulas of SURVEY.md Appendix A
(the reference relies on autograd).  Both directions are pinned against
outputs of the real reference in ``tests/golden/*.npz`` (generated here by
``oracle/gen_golden.py``, which imports ``/root/reference``).

Parity status: PINNED (tests/test_oracle_golden.py).

Complex tensors are passed as two planar arrays ``(re, im)`` of equal shape,
which is the reference's own ``Cplx`` layout (cplxmodule/cplx.py:10-52).
All functions are dtype-generic: the arithmetic runs in the dtype of the
inputs (float32 or float64).
"""
import numpy as np
from scipy.special import expi as _scipy_expi

EULER_GAMMA = float(np.euler_gamma)

# constants of the softplus-sigmoid KL approximation,
# cplxmodule/nn/relevance/real/vd.py:74-76
K1, K2, K3 = 0.63576, 1.87320, 1.48695

KINDS = ("real_vd", "real_ard", "cplx_vd", "cplx_ard")
EXT_KINDS = ("cplx_vd_approx", "cplx_vd_scalefree", "cplx_vd_bogus")   # nn/relevance/extensions/complex.py


# --------------------------------------------------------------------------- #
#  complex linear                                                             #
# --------------------------------------------------------------------------- #
def cplx_linear(xr, xi, wr, wi, br=None, bi=None, algo="4m"):
    """y = x W^T + b for planar complex tensors.

    4m: cplxmodule/cplx.py:634-648 (linear_naive);
    3m: cplxmodule/cplx.py:669-694 (linear_3m, Gauss trick);
    cat: cplxmodule/cplx.py:651-666 (one fat real GEMM).
    """
    if algo == "4m":
        re = xr @ wr.T - xi @ wi.T
        im = xr @ wi.T + xi @ wr.T
    elif algo == "3m":
        k1 = (xr + xi) @ wr.T
        k2 = xr @ (wi - wr).T
        k3 = xi @ (wr + wi).T
        re, im = k1 - k3, k1 + k2
    elif algo == "cat":
        ww = np.concatenate(
            [np.concatenate([wr, wi], 0), np.concatenate([-wi, wr], 0)], 1)
        out = np.concatenate([xr, xi], -1) @ ww.T
        re, im = np.split(out, 2, axis=-1)
    else:
        raise ValueError(algo)
    if br is not None:
        re, im = re + br, im + bi
    return re, im


def cplx_linear_bwd(gr, gi, xr, xi, wr, wi, has_bias=True):
    """Gradients of ``cplx_linear``: dX = G conj(W),
    dW = G^T conj(X), db = sum_b G.  Inputs may carry leading batch dims."""
    O, I = wr.shape
    g2r, g2i = gr.reshape(-1, O), gi.reshape(-1, O)
    x2r, x2i = xr.reshape(-1, I), xi.reshape(-1, I)
    dxr = (g2r @ wr + g2i @ wi).reshape(xr.shape)
    dxi = (-g2r @ wi + g2i @ wr).reshape(xr.shape)
    dwr = g2r.T @ x2r + g2i.T @ x2i
    dwi = -g2r.T @ x2i + g2i.T @ x2r
    out = dict(dxr=dxr, dxi=dxi, dwr=dwr, dwi=dwi)
    if has_bias:
        out["dbr"], out["dbi"] = g2r.sum(0), g2i.sum(0)
    return out


def cplx_matmul(ur, ui, vr, vi):
    """Cplx.__matmul__, cplxmodule/cplx.py:167-174."""
    return ur @ vr - ui @ vi, ui @ vr + ur @ vi


# --------------------------------------------------------------------------- #
#  noise                                                                      #
# --------------------------------------------------------------------------- #
def cplx_randn_from_tape(tape):
    """cplx.randn, cplxmodule/cplx.py:544-550: ONE normal draw of shape
    [2, *size] divided by sqrt(2); plane 0 -> real, plane 1 -> imag."""
    z = tape / np.asarray(np.sqrt(2.0), dtype=tape.dtype)
    return z[0], z[1]


# --------------------------------------------------------------------------- #
#  local reparameterization                                                   #
# --------------------------------------------------------------------------- #
def lrt_cplx_linear(xr, xi, wr, wi, br, bi, log_sigma2, eps_r, eps_i):
    """CplxLinearGaussian.forward in training mode,
    cplxmodule/nn/relevance/complex/base.py:43-56."""
    mur, mui = cplx_linear(xr, xi, wr, wi, br, bi)
    s2 = (xr * xr + xi * xi) @ np.exp(log_sigma2).T
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mur + eps_r * sd, mui + eps_i * sd, dict(mur=mur, mui=mui, s2=s2)


def lrt_cplx_linear_bwd(gr, gi, xr, xi, wr, wi, log_sigma2, eps_r, eps_i,
                        has_bias=True):
    """SURVEY.md A.2 (complex).  torch.clamp passes the gradient AT the
    boundary (s2 == 1e-8) and blocks it below."""
    dt = xr.dtype
    S = np.exp(log_sigma2)
    a = xr * xr + xi * xi
    s2 = a @ S.T
    lo = np.asarray(1e-8, dt)
    sd = np.sqrt(np.maximum(s2, lo))
    gsd = gr * eps_r + gi * eps_i
    gs2 = np.where(s2 >= lo, gsd * np.asarray(0.5, dt) / sd, np.asarray(0, dt))
    ga = gs2 @ S
    out = cplx_linear_bwd(gr, gi, xr, xi, wr, wi, has_bias)
    out["dxr"] = out["dxr"] + 2 * xr * ga
    out["dxi"] = out["dxi"] + 2 * xi * ga
    out["dlog_sigma2"] = (gs2.reshape(-1, S.shape[0]).T
                          @ a.reshape(-1, S.shape[1])) * S
    return out


def lrt_real_linear(x, w, b, log_sigma2, eps):
    """LinearGaussian.forward in training mode,
    cplxmodule/nn/relevance/real/base.py:43-49."""
    mu = x @ w.T
    if b is not None:
        mu = mu + b
    s2 = (x * x) @ np.exp(log_sigma2).T
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mu + eps * sd, dict(mu=mu, s2=s2)


def lrt_real_linear_bwd(g, x, w, log_sigma2, eps, has_bias=True):
    """SURVEY.md A.2 (real)."""
    dt = x.dtype
    S = np.exp(log_sigma2)
    a = x * x
    s2 = a @ S.T
    lo = np.asarray(1e-8, dt)
    sd = np.sqrt(np.maximum(s2, lo))
    gs2 = np.where(s2 >= lo, g * eps * np.asarray(0.5, dt) / sd,
                   np.asarray(0, dt))
    ga = gs2 @ S
    O, I = w.shape
    g2, x2 = g.reshape(-1, O), x.reshape(-1, I)
    out = dict(dx=(g2 @ w).reshape(x.shape) + 2 * x * ga, dw=g2.T @ x2,
               dlog_sigma2=(gs2.reshape(-1, O).T @ a.reshape(-1, I)) * S)
    if has_bias:
        out["db"] = g2.sum(0)
    return out


# --------------------------------------------------------------------------- #
#  transposed convolution                                                     #
# --------------------------------------------------------------------------- #
def _convt_out_shape(x_shape, w_shape, stride, padding, output_padding, dilation, groups):
    (sh, sw), (ph, pw), (oh, ow), (dh, dw) = (_pair(v) for v in (stride, padding, output_padding, dilation))
    B, _, H, W = x_shape
    _, Cog, kh, kw = w_shape
    return (B, Cog * groups, (H - 1) * sh - 2 * ph + dh * (kh - 1) + oh + 1,
            (W - 1) * sw - 2 * pw + dw * (kw - 1) + ow + 1)


def real_conv_transpose2d(x, w, stride=1, padding=0, output_padding=0, dilation=1, groups=1):
    """torch.nn.functional.conv_transpose2d = the adjoint (input gradient) of conv2d with the same
    weight, evaluated at x; weight [Cin, Cout / groups, kh, kw]."""
    yshape = _convt_out_shape(x.shape, w.shape, stride, padding, output_padding, dilation, groups)
    return real_conv2d_bwd(x, np.zeros(yshape, x.dtype), w, stride, padding, dilation, groups)[0]


def cplx_conv_transpose2d(xr, xi, wr, wi, br=None, bi=None, stride=1, padding=0, output_padding=0,
                          dilation=1, groups=1):
    """cplx.conv_transposend_naive + bias, cplxmodule/cplx.py:860-873, 937-943 (no conjugation)."""
    t = lambda x, w: real_conv_transpose2d(x, w, stride, padding, output_padding, dilation, groups)  # noqa: E731
    re, im = t(xr, wr) - t(xi, wi), t(xr, wi) + t(xi, wr)
    if br is not None:
        re, im = re + br.reshape(-1, 1, 1), im + bi.reshape(-1, 1, 1)
    return re, im


def cplx_conv_transpose2d_bwd(gr, gi, xr, xi, wr, wi, stride=1, padding=0, output_padding=0, dilation=1,
                              groups=1):
    """T_w is the adjoint of Conv_w, so d/dx is Conv_w of the output gradient and d/dw is the conv
    weight gradient with the roles (output gradient, input) = (x, g)."""
    c = lambda g, w: real_conv2d(g, w, stride, padding, dilation, groups)  # noqa: E731
    wg = lambda x, g: real_conv2d_bwd(x, g, wr, stride, padding, dilation, groups)[1]  # noqa: E731
    return dict(dxr=c(gr, wr) + c(gi, wi), dxi=c(gi, wr) - c(gr, wi),
                dwr=wg(xr, gr) + wg(xi, gi), dwi=wg(xr, gi) - wg(xi, gr),
                dbr=gr.sum((0, 2, 3)), dbi=gi.sum((0, 2, 3)))


# --------------------------------------------------------------------------- #
#  3-d convolution / pooling                                                  #
# --------------------------------------------------------------------------- #
def _triple(v):
    return (v, v, v) if isinstance(v, (int, np.integer)) else tuple(v)


def _conv3d_taps(x_shape, w_shape, stride, padding, dilation):
    (sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = _triple(stride), _triple(padding), _triple(dilation)
    kd, kh, kw = w_shape[2:]
    D, H, W = x_shape[2:]
    Do = (D + 2 * pd - dd * (kd - 1) - 1) // sd + 1
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    taps = []
    for a in range(kd):
        for i in range(kh):
            for j in range(kw):
                taps.append(((a, i, j), (slice(a * dd, a * dd + sd * (Do - 1) + 1, sd),
                                         slice(i * dh, i * dh + sh * (Ho - 1) + 1, sh),
                                         slice(j * dw, j * dw + sw * (Wo - 1) + 1, sw))))
    return (pd, ph, pw), (Do, Ho, Wo), taps


def real_conv3d(x, w, stride=1, padding=0, dilation=1, groups=1):
    """torch.nn.functional.conv3d (cross-correlation, zero padding), one einsum per kernel tap."""
    (pd, ph, pw), (Do, Ho, Wo), taps = _conv3d_taps(x.shape, w.shape, stride, padding, dilation)
    B, C = x.shape[:2]
    Co, Cg = w.shape[:2]
    xp = np.pad(x, ((0, 0), (0, 0), (pd, pd), (ph, ph), (pw, pw)))
    out = np.zeros((B, groups, Co // groups, Do, Ho, Wo), x.dtype)
    wg = w.reshape(groups, Co // groups, Cg, *w.shape[2:])
    for (a, i, j), (zs, ys, xs) in taps:
        v = xp[:, :, zs, ys, xs].reshape(B, groups, Cg, Do, Ho, Wo)
        out += np.einsum("goc,bgcdhw->bgodhw", wg[:, :, :, a, i, j], v)
    return out.reshape(B, Co, Do, Ho, Wo)


def real_conv3d_bwd(g, x, w, stride=1, padding=0, dilation=1, groups=1):
    """(dx, dw) of `real_conv3d`: the adjoint of every tap."""
    (pd, ph, pw), (Do, Ho, Wo), taps = _conv3d_taps(x.shape, w.shape, stride, padding, dilation)
    B, C, D, H, W = x.shape
    Co, Cg = w.shape[:2]
    xp = np.pad(x, ((0, 0), (0, 0), (pd, pd), (ph, ph), (pw, pw)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w).reshape(groups, Co // groups, Cg, *w.shape[2:])
    wg = w.reshape(dw.shape)
    gg = g.reshape(B, groups, Co // groups, Do, Ho, Wo)
    for (a, i, j), (zs, ys, xs) in taps:
        v = xp[:, :, zs, ys, xs].reshape(B, groups, Cg, Do, Ho, Wo)
        dw[:, :, :, a, i, j] = np.einsum("bgodhw,bgcdhw->goc", gg, v)
        dxp[:, :, zs, ys, xs] += np.einsum("goc,bgodhw->bgcdhw", wg[:, :, :, a, i, j], gg).reshape(B, C, Do, Ho, Wo)
    return dxp[:, :, pd: pd + D, ph: ph + H, pw: pw + W], dw.reshape(w.shape)


def circular_pad3d(x, padding):
    """symmetric_circular_padding for 5-d input (cplxmodule/cplx.py:699-712): as in 2-d, the FIRST
    entry of `padding` wraps the LAST dim."""
    (wl, wr_), (hl, hr), (dl, dr) = [((p + 1) // 2, p // 2) for p in _triple(padding)]
    return np.pad(x, ((0, 0), (0, 0), (dl, dr), (hl, hr), (wl, wr_)), mode="wrap")


def cplx_conv3d(xr, xi, wr, wi, br=None, bi=None, stride=1, padding=0, dilation=1, groups=1,
                padding_mode="zeros"):
    """cplx.conv3d -> convnd, cplxmodule/cplx.py:770-800, 841-857."""
    if padding_mode == "circular":
        xr, xi, padding = circular_pad3d(xr, padding), circular_pad3d(xi, padding), 0
    f = lambda x, w: real_conv3d(x, w, stride, padding, dilation, groups)  # noqa: E731
    re, im = f(xr, wr) - f(xi, wi), f(xr, wi) + f(xi, wr)
    if br is not None:
        re, im = re + br.reshape(-1, 1, 1, 1), im + bi.reshape(-1, 1, 1, 1)
    return re, im


def cplx_conv3d_bwd(gr, gi, xr, xi, wr, wi, stride=1, padding=0, dilation=1, groups=1):
    """dX = G (*) conj(W), dW = G (*) conj(X) (zeros padding), db = sum G."""
    b = lambda g, x, w: real_conv3d_bwd(g, x, w, stride, padding, dilation, groups)  # noqa: E731
    dx_rr, dw_rr = b(gr, xr, wr)
    dx_ii, dw_ii = b(gi, xi, wr)       # wrt (xi, wr) through the imaginary output
    dx_ri, dw_ri = b(gr, xi, wi)       # re -= conv(xi, wi)
    dx_ir, dw_ir = b(gi, xr, wi)       # im += conv(xr, wi)
    return dict(dxr=dx_rr + dx_ir, dxi=dx_ii - dx_ri, dwr=dw_rr + dw_ii, dwi=dw_ir - dw_ri,
                dbr=gr.sum((0, 2, 3, 4)), dbi=gi.sum((0, 2, 3, 4)))


def lrt_cplx_conv3d(xr, xi, wr, wi, br, bi, log_sigma2, eps_r, eps_i, **kw):
    """CplxConvNdGaussianMixin._forward_impl with F.conv3d, complex/base.py:120-135."""
    mur, mui = cplx_conv3d(xr, xi, wr, wi, br, bi, **kw)
    s2 = real_conv3d(xr * xr + xi * xi, np.exp(log_sigma2), **kw)
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mur + eps_r * sd, mui + eps_i * sd, dict(mur=mur, mui=mui, s2=s2)


def lrt_cplx_conv3d_bwd(gr, gi, xr, xi, wr, wi, log_sigma2, eps_r, eps_i, **kw):
    dt = xr.dtype
    S = np.exp(log_sigma2)
    a = xr * xr + xi * xi
    s2 = real_conv3d(a, S, **kw)
    lo = np.asarray(1e-8, dt)
    sd = np.sqrt(np.maximum(s2, lo))
    gs2 = np.where(s2 >= lo, (gr * eps_r + gi * eps_i) * np.asarray(0.5, dt) / sd, np.asarray(0, dt))
    out = cplx_conv3d_bwd(gr, gi, xr, xi, wr, wi, **kw)
    ga, dS = real_conv3d_bwd(gs2, a, S, **kw)
    out["dxr"] = out["dxr"] + 2 * xr * ga
    out["dxi"] = out["dxi"] + 2 * xi * ga
    out["dlog_sigma2"] = dS * S
    return out


def cplx_max_pool3d(zr, zi, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """cplx.max_pool3d (cplxmodule/cplx.py:1114-1175, 1193-1200): direct 3-d window scan, first
    maximum of |z| in (d, h, w) order.  Returns (yr, yi, flat index into D*H*W)."""
    k, p, dl = _triple(kernel_size), _triple(padding), _triple(dilation)
    s = k if stride is None else _triple(stride)
    B, C, D, H, W = zr.shape
    osz = [_pool_out(L, k[n], s[n], p[n], dl[n], ceil_mode) for n, L in enumerate((D, H, W))]
    mod = cplx_abs(zr, zi)
    idx = np.zeros((B, C, *osz), np.int64)
    for od in range(osz[0]):
        for oh in range(osz[1]):
            for ow in range(osz[2]):
                best = np.full((B, C), -np.inf)
                sel = np.full((B, C), -1, np.int64)
                for a in range(k[0]):
                    d = od * s[0] - p[0] + a * dl[0]
                    for i in range(k[1]):
                        h = oh * s[1] - p[1] + i * dl[1]
                        for j in range(k[2]):
                            w = ow * s[2] - p[2] + j * dl[2]
                            if not (0 <= d < D and 0 <= h < H and 0 <= w < W):
                                continue
                            m = mod[:, :, d, h, w]
                            take = (sel < 0) | (m > best)
                            best = np.where(take, m, best)
                            sel = np.where(take, (d * H + h) * W + w, sel)
                idx[:, :, od, oh, ow] = sel
    fi = idx.reshape(B, C, -1)
    yr = np.take_along_axis(zr.reshape(B, C, -1), fi, -1).reshape(idx.shape)
    yi = np.take_along_axis(zi.reshape(B, C, -1), fi, -1).reshape(idx.shape)
    return yr, yi, idx


# --------------------------------------------------------------------------- #
#  bilinear layers                                                            #
# --------------------------------------------------------------------------- #
def real_bilinear(x1, x2, w, b=None):
    """torch.nn.functional.bilinear: y[b,o] = sum_ij x1[b,i] w[o,i,j] x2[b,j] (+ b[o])."""
    y = np.einsum("bi,oij,bj->bo", x1, w, x2)
    return y if b is None else y + b


def cplx_bilinear(x1r, x1i, x2r, x2i, wr, wi, br=None, bi=None, conjugate=True):
    """cplx.bilinear_naive, cplxmodule/cplx.py:1062-1087: four real bilinear forms per weight
    plane, combined as conj?(x1) (x) x2 = P + iQ,  y = (P Wr - Q Wi) + i (P Wi + Q Wr)."""
    f = real_bilinear
    au_r, au_i = f(x1r, x2r, wr), f(x1r, x2r, wi)
    av_r, av_i = f(x1r, x2i, wr), f(x1r, x2i, wi)
    bu_r, bu_i = f(x1i, x2r, wr), f(x1i, x2r, wi)
    bv_r, bv_i = f(x1i, x2i, wr), f(x1i, x2i, wi)
    if conjugate:
        pp_r, pp_i, qq_r, qq_i = au_r + bv_r, au_i + bv_i, av_r - bu_r, av_i - bu_i
    else:
        pp_r, pp_i, qq_r, qq_i = au_r - bv_r, au_i - bv_i, av_r + bu_r, av_i + bu_i
    re, im = pp_r - qq_i, pp_i + qq_r
    if br is not None:
        re, im = re + br, im + bi
    return re, im


def cplx_bilinear_bwd(gr, gi, x1r, x1i, x2r, x2i, wr, wi, conjugate=True, has_bias=True):
    """Gradients of `cplx_bilinear` for a real loss (planar convention of SURVEY A.1:
    d(u v) -> du = g conj(v)).  With u = conj?(x1):  y_o = sum_ij u_i W_oij x2_j."""
    g = gr + 1j * gi
    x1 = x1r + 1j * x1i
    u = np.conj(x1) if conjugate else x1
    x2 = x2r + 1j * x2i
    w = wr + 1j * wi
    du = np.einsum("bo,oij,bj->bi", g, np.conj(w), np.conj(x2))
    dx1 = np.conj(du) if conjugate else du
    dx2 = np.einsum("bo,oij,bi->bj", g, np.conj(w), np.conj(u))
    dw = np.einsum("bo,bi,bj->oij", g, np.conj(u), np.conj(x2))
    dt = x1r.dtype
    out = dict(dx1r=dx1.real.astype(dt), dx1i=dx1.imag.astype(dt), dx2r=dx2.real.astype(dt),
               dx2i=dx2.imag.astype(dt), dwr=dw.real.astype(dt), dwi=dw.imag.astype(dt))
    if has_bias:
        out["dbr"], out["dbi"] = gr.sum(0), gi.sum(0)
    return out


def lrt_cplx_bilinear(x1r, x1i, x2r, x2i, wr, wi, br, bi, log_sigma2, eps_r, eps_i, conjugate=True):
    """CplxBilinearGaussian.forward in training mode,
    cplxmodule/nn/relevance/complex/base.py:73-84."""
    mur, mui = cplx_bilinear(x1r, x1i, x2r, x2i, wr, wi, br, bi, conjugate)
    s2 = real_bilinear(x1r * x1r + x1i * x1i, x2r * x2r + x2i * x2i, np.exp(log_sigma2))
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mur + eps_r * sd, mui + eps_i * sd, dict(mur=mur, mui=mui, s2=s2)


def lrt_cplx_bilinear_bwd(gr, gi, x1r, x1i, x2r, x2i, wr, wi, log_sigma2, eps_r, eps_i,
                          conjugate=True, has_bias=True):
    """SURVEY A.2 with the bilinear variance s2 = sum_ij a1_i S_oij a2_j."""
    dt = x1r.dtype
    S = np.exp(log_sigma2)
    a1, a2 = x1r * x1r + x1i * x1i, x2r * x2r + x2i * x2i
    s2 = real_bilinear(a1, a2, S)
    lo = np.asarray(1e-8, dt)
    sd = np.sqrt(np.maximum(s2, lo))
    gs2 = np.where(s2 >= lo, (gr * eps_r + gi * eps_i) * np.asarray(0.5, dt) / sd, np.asarray(0, dt))
    out = cplx_bilinear_bwd(gr, gi, x1r, x1i, x2r, x2i, wr, wi, conjugate, has_bias)
    ga1 = np.einsum("bo,oij,bj->bi", gs2, S, a2)
    ga2 = np.einsum("bo,oij,bi->bj", gs2, S, a1)
    out["dx1r"] = out["dx1r"] + 2 * x1r * ga1
    out["dx1i"] = out["dx1i"] + 2 * x1i * ga1
    out["dx2r"] = out["dx2r"] + 2 * x2r * ga2
    out["dx2i"] = out["dx2i"] + 2 * x2i * ga2
    out["dlog_sigma2"] = np.einsum("bo,bi,bj->oij", gs2, a1, a2) * S
    return out


def lrt_real_bilinear(x1, x2, w, b, log_sigma2, eps):
    """BilinearGaussian.forward in training mode, cplxmodule/nn/relevance/real/base.py:66-77."""
    mu = real_bilinear(x1, x2, w, b)
    s2 = real_bilinear(x1 * x1, x2 * x2, np.exp(log_sigma2))
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mu + eps * sd, dict(mu=mu, s2=s2)


def lrt_real_bilinear_bwd(g, x1, x2, w, log_sigma2, eps, has_bias=True, noise=True):
    """Gradients of `lrt_real_bilinear` (noise=False: of `real_bilinear` alone)."""
    dt = x1.dtype
    out = dict(dx1=np.einsum("bo,oij,bj->bi", g, w, x2), dx2=np.einsum("bo,oij,bi->bj", g, w, x1),
               dw=np.einsum("bo,bi,bj->oij", g, x1, x2))
    if has_bias:
        out["db"] = g.sum(0)
    if noise:
        S = np.exp(log_sigma2)
        a1, a2 = x1 * x1, x2 * x2
        s2 = real_bilinear(a1, a2, S)
        lo = np.asarray(1e-8, dt)
        sd = np.sqrt(np.maximum(s2, lo))
        gs2 = np.where(s2 >= lo, g * eps * np.asarray(0.5, dt) / sd, np.asarray(0, dt))
        out["dx1"] = out["dx1"] + 2 * x1 * np.einsum("bo,oij,bj->bi", gs2, S, a2)
        out["dx2"] = out["dx2"] + 2 * x2 * np.einsum("bo,oij,bi->bj", gs2, S, a1)
        out["dlog_sigma2"] = np.einsum("bo,bi,bj->oij", gs2, a1, a2) * S
    return out


# --------------------------------------------------------------------------- #
#  log-alpha, KL penalties, masks                                             #
# --------------------------------------------------------------------------- #
def cplx_abs(wr, wi):
    """Cplx.__abs__, cplxmodule/cplx.py:183-192 (stack + 2-norm over dim 0).
    torch's CPU kernel evaluates sqrt(fma(wi, wi, round(wr*wr))): verified
    bit-for-bit on 2^20 float32 samples ("mask exactness")."""
    if wr.dtype == np.float32:
        sq = (wr * wr).astype(np.float64) + wi.astype(np.float64) ** 2
        return np.sqrt(sq.astype(np.float32))
    return np.sqrt(wr * wr + wi * wi)


def cplx_abs_bwd(g, zr, zi):
    """Gradient of abs(Cplx) = norm(stack([re, im]), dim=0) (cplx.py:183-192): g z / |z|, and the 2-norm's
    subgradient 0 at z == 0."""
    r = np.sqrt(zr * zr + zi * zi)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(r > 0, g / r, 0.0)
    return s * zr, s * zi


def log_alpha_bwd(g, wr, wi=None):
    """d/dw of log_sigma2 - 2 log(abs(w) + 1e-12) (complex/base.py:27-31, real/base.py:23-26) times g;
    the gradient wrt log_sigma2 is g itself."""
    if wi is None:
        return -2.0 * g * np.sign(wr) / (np.abs(wr) + 1e-12), None
    th = np.sqrt(wr * wr + wi * wi)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(th > 0, -2.0 * g / (th * (th + 1e-12)), 0.0)
    return c * wr, c * wi


def binarize_masks(state_dict, masks):
    """nn/masked/base.py:232-264: weights times their (soft) mask with -0.0 cleaned up; masks -> 0/1."""
    out = {}
    for name, par in state_dict.items():
        if "weight" in name:
            key = name.rsplit("weight", 1)[0] + "mask"
            if key in masks:
                par = par * masks[key].astype(par.dtype)
                par = np.where(np.abs(par) == 0, np.zeros_like(par), par)
        out[name] = par
    return out, {k: (m != 0).astype(m.dtype) for k, m in masks.items()}


def log_alpha(log_sigma2, wr, wi=None):
    """GaussianMixin.log_alpha: complex cplxmodule/nn/relevance/complex/base.py:27-31,
    real cplxmodule/nn/relevance/real/base.py:23-26."""
    theta = np.abs(wr) if wi is None else cplx_abs(wr, wi)
    eps0 = np.asarray(1e-12, log_sigma2.dtype)
    return log_sigma2 - 2 * np.log(theta + eps0)


def softplus(x):
    """torch.nn.functional.softplus, beta=1, threshold=20."""
    with np.errstate(over="ignore"):
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def softplus_grad(x):
    """torch's softplus backward: exactly 1 above the linear threshold 20."""
    return np.where(x > 20, np.asarray(1, x.dtype), sigmoid(x))


def expi(x):
    """torch_expi == scipy.special.expi evaluated in the input dtype's loop
    (float32 loop computes in double, rounds once),
    cplxmodule/nn/relevance/complex/vd.py:31-36."""
    return _scipy_expi(x).astype(x.dtype)


def penalty(kind, log_sigma2, wr, wi=None):
    """Elementwise KL penalty, shape of the weight.
    real_vd  cplxmodule/nn/relevance/real/vd.py:54-76
    real_ard cplxmodule/nn/relevance/real/ard.py:10-39
    cplx_vd  cplxmodule/nn/relevance/complex/vd.py:95-99
    cplx_ard cplxmodule/nn/relevance/complex/ard.py:9-39
    """
    dt = log_sigma2.dtype
    t = -log_alpha(log_sigma2, wr, wi)
    if kind == "real_vd":
        sg = sigmoid(np.asarray(K3, dt) * t - np.asarray(K2, dt))
        return softplus(t) / 2 + np.asarray(K1, dt) * sg
    if kind == "real_ard":
        return np.asarray(0.5, dt) * softplus(t)
    if kind == "cplx_vd":
        with np.errstate(over="ignore"):
            return np.asarray(EULER_GAMMA, dt) + t - expi(-np.exp(t))
    if kind == "cplx_ard":
        return softplus(t)
    if kind == "cplx_vd_approx":      # extensions/complex.py:113-117
        return softplus(t) + np.asarray(0.57810, dt) * sigmoid(np.asarray(1.36526, dt) * t - np.asarray(1.45926, dt))
    if kind == "cplx_vd_scalefree":   # extensions/complex.py:43-46 (t = 2 log|w| - log_sigma2)
        log_abs_w = (t + log_sigma2) / 2
        with np.errstate(over="ignore"):
            return log_abs_w - log_sigma2 - np.asarray(0.5, dt) * expi(-np.exp(t))
    if kind == "cplx_vd_bogus":       # extensions/complex.py:142-160: -log_alpha - 0 (forward of Ei dropped)
        return t
    raise ValueError(kind)


def penalty_dt(kind, t):
    """f'(t) with t = -log_alpha."""
    dt = t.dtype
    if kind == "real_vd":
        u = np.asarray(K3, dt) * t - np.asarray(K2, dt)
        su = sigmoid(u)
        return softplus_grad(t) / 2 + np.asarray(K1 * K3, dt) * su * (1 - su)
    if kind == "real_ard":
        return softplus_grad(t) / 2
    if kind in ("cplx_vd", "cplx_vd_bogus"):
        with np.errstate(over="ignore"):
            return -np.expm1(-np.exp(t))
    if kind == "cplx_ard":
        return softplus_grad(t)
    if kind == "cplx_vd_approx":
        su = sigmoid(np.asarray(1.36526, dt) * t - np.asarray(1.45926, dt))
        return softplus_grad(t) + np.asarray(0.57810 * 1.36526, dt) * su * (1 - su)
    if kind == "cplx_vd_scalefree":   # the part through t; the direct -ls2/2 term is in penalty_bwd
        with np.errstate(over="ignore"):
            return -np.expm1(-np.exp(t)) / 2
    raise ValueError(kind)


def penalty_bwd(kind, g, log_sigma2, wr, wi=None):
    """Gradient of sum(g * penalty) wrt (log_sigma2, wr, wi), SURVEY.md A.3.
    The gradient wrt the weight is 0 where |w| == 0 (torch subgradient)."""
    dt = log_sigma2.dtype
    t = -log_alpha(log_sigma2, wr, wi)
    fp = g * penalty_dt(kind, t)
    eps0 = np.asarray(1e-12, dt)
    out = dict(dlog_sigma2=-fp)
    if kind == "cplx_vd_scalefree":
        out["dlog_sigma2"] = -fp - g * np.asarray(0.5, dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        if wi is None:
            theta = np.abs(wr)
            out["dwr"] = np.where(theta > 0,
                                  fp * 2 * np.sign(wr) / (theta + eps0), 0
                                  ).astype(dt)
        else:
            theta = cplx_abs(wr, wi)
            den = theta * (theta + eps0)
            out["dwr"] = np.where(theta > 0, fp * 2 * wr / den, 0).astype(dt)
            out["dwi"] = np.where(theta > 0, fp * 2 * wi / den, 0).astype(dt)
    return out


def relevance_mask(threshold, log_sigma2, wr, wi=None):
    """RelevanceMixin.relevance: real cplxmodule/nn/relevance/real/vd.py:16-19,
    complex cplxmodule/nn/relevance/complex/vd.py:50-53.  Float 0/1 mask."""
    la = log_alpha(log_sigma2, wr, wi)
    return (la <= np.asarray(threshold, la.dtype)).astype(la.dtype)


# --------------------------------------------------------------------------- #
#  convolution                                                                #
# --------------------------------------------------------------------------- #
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _im2col(x, kh, kw, stride, padding, dilation):
    """x [B, C, H, W] -> cols [B, C, kh, kw, Ho, Wo] with zero padding."""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    cols = np.empty((B, C, kh, kw, Ho, Wo), x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, i, j] = xp[:, :, i * dh: i * dh + sh * (Ho - 1) + 1: sh,
                                  j * dw: j * dw + sw * (Wo - 1) + 1: sw]
    return cols


def real_conv2d(x, w, stride=1, padding=0, dilation=1, groups=1):
    """torch.nn.functional.conv2d (cross-correlation), zero padding."""
    Co, Cg, kh, kw = w.shape
    cols = _im2col(x, kh, kw, stride, padding, dilation)
    B, C, _, _, Ho, Wo = cols.shape
    cols = cols.reshape(B, groups, Cg * kh * kw, Ho * Wo)
    wg = w.reshape(groups, Co // groups, Cg * kh * kw)
    out = np.einsum("gok,bgkp->bgop", wg, cols)
    return out.reshape(B, Co, Ho, Wo)


def real_conv2d_bwd(g, x, w, stride=1, padding=0, dilation=1, groups=1):
    """(dx, dw) of ``real_conv2d`` via col2im."""
    (sh, sw), (ph, pw), (dh, dw_) = _pair(stride), _pair(padding), _pair(dilation)
    Co, Cg, kh, kw = w.shape
    B, C, H, W = x.shape
    cols = _im2col(x, kh, kw, stride, padding, dilation)
    Ho, Wo = cols.shape[-2:]
    colsg = cols.reshape(B, groups, Cg * kh * kw, Ho * Wo)
    gg = g.reshape(B, groups, Co // groups, Ho * Wo)
    wg = w.reshape(groups, Co // groups, Cg * kh * kw)
    dw = np.einsum("bgop,bgkp->gok", gg, colsg).reshape(w.shape)
    dcols = np.einsum("gok,bgop->bgkp", wg, gg).reshape(B, C, kh, kw, Ho, Wo)
    dxp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), x.dtype)
    for i in range(kh):
        for j in range(kw):
            dxp[:, :, i * dh: i * dh + sh * (Ho - 1) + 1: sh,
                j * dw_: j * dw_ + sw * (Wo - 1) + 1: sw] += dcols[:, :, i, j]
    return dxp[:, :, ph: ph + H, pw: pw + W], dw


def circular_pad2d(x, padding):
    """symmetric_circular_padding, cplxmodule/cplx.py:701-714: each spatial dim
    gets ((pad+1)//2, pad//2); F.pad's tuple runs from the LAST dim backwards,
    so the first entry of ``padding`` pads the last dim."""
    pads = _pair(padding)
    (wl, wr_), (hl, hr) = [((p + 1) // 2, p // 2) for p in pads]
    return np.pad(x, ((0, 0), (0, 0), (hl, hr), (wl, wr_)), mode="wrap")


def cplx_conv2d(xr, xi, wr, wi, br=None, bi=None, stride=1, padding=0,
                dilation=1, groups=1, padding_mode="zeros"):
    """cplx.conv2d -> convnd, cplxmodule/cplx.py:770-800, 822-838:
    re = conv(xr, wr) - conv(xi, wi); im = conv(xr, wi) + conv(xi, wr)
    (no conjugation), bias broadcast over the spatial dims."""
    if padding_mode == "circular":
        xr, xi = circular_pad2d(xr, padding), circular_pad2d(xi, padding)
        padding = 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    a = (stride, padding, dilation, groups)
    re = real_conv2d(xr, wr, *a) - real_conv2d(xi, wi, *a)
    im = real_conv2d(xr, wi, *a) + real_conv2d(xi, wr, *a)
    if br is not None:
        re, im = re + br.reshape(1, -1, 1, 1), im + bi.reshape(1, -1, 1, 1)
    return re, im


def cplx_conv2d_bwd(gr, gi, xr, xi, wr, wi, stride=1, padding=0, dilation=1,
                    groups=1, has_bias=True):
    """Zero-padding complex conv gradients: the A.1 algebra with conv."""
    a = (stride, padding, dilation, groups)
    # re = c(xr,wr) - c(xi,wi); im = c(xr,wi) + c(xi,wr)
    dx_rr, dw_rr = real_conv2d_bwd(gr, xr, wr, *a)   # d re / (xr, wr)
    dx_ii, dw_ii = real_conv2d_bwd(gr, xi, wi, *a)   # d(-re) / (xi, wi)
    dx_ri, dw_ri = real_conv2d_bwd(gi, xr, wi, *a)   # d im / (xr, wi)
    dx_ir, dw_ir = real_conv2d_bwd(gi, xi, wr, *a)   # d im / (xi, wr)
    out = dict(dxr=dx_rr + dx_ri, dxi=-dx_ii + dx_ir,
               dwr=dw_rr + dw_ir, dwi=-dw_ii + dw_ri)
    if has_bias:
        out["dbr"], out["dbi"] = gr.sum((0, 2, 3)), gi.sum((0, 2, 3))
    return out


def lrt_cplx_conv2d(xr, xi, wr, wi, br, bi, log_sigma2, eps_r, eps_i,
                    stride=1, padding=0, dilation=1, groups=1):
    """CplxConvNdGaussianMixin._forward_impl,
    cplxmodule/nn/relevance/complex/base.py:120-135 (zeros padding only)."""
    a = (stride, padding, dilation, groups)
    mur, mui = cplx_conv2d(xr, xi, wr, wi, br, bi, *a)
    s2 = real_conv2d(xr * xr + xi * xi, np.exp(log_sigma2), *a)
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mur + eps_r * sd, mui + eps_i * sd, dict(mur=mur, mui=mui, s2=s2)


def lrt_cplx_conv2d_bwd(gr, gi, xr, xi, wr, wi, log_sigma2, eps_r, eps_i,
                        stride=1, padding=0, dilation=1, groups=1,
                        has_bias=True):
    dt = xr.dtype
    a = (stride, padding, dilation, groups)
    S = np.exp(log_sigma2)
    ab = xr * xr + xi * xi
    s2 = real_conv2d(ab, S, *a)
    lo = np.asarray(1e-8, dt)
    sd = np.sqrt(np.maximum(s2, lo))
    gs2 = np.where(s2 >= lo, (gr * eps_r + gi * eps_i) * np.asarray(0.5, dt) / sd,
                   np.asarray(0, dt))
    ga, dS = real_conv2d_bwd(gs2, ab, S, *a)
    out = cplx_conv2d_bwd(gr, gi, xr, xi, wr, wi, *a, has_bias=has_bias)
    out["dxr"] = out["dxr"] + 2 * xr * ga
    out["dxi"] = out["dxi"] + 2 * xi * ga
    out["dlog_sigma2"] = dS * S
    return out


def lrt_real_conv2d(x, w, b, log_sigma2, eps, stride=1, padding=0, dilation=1,
                    groups=1):
    """ConvNdGaussianMixin._forward_impl,
    cplxmodule/nn/relevance/real/base.py:116-163."""
    a = (stride, padding, dilation, groups)
    mu = real_conv2d(x, w, *a)
    if b is not None:
        mu = mu + b.reshape(1, -1, 1, 1)
    s2 = real_conv2d(x * x, np.exp(log_sigma2), *a)
    sd = np.sqrt(np.maximum(s2, np.asarray(1e-8, s2.dtype)))
    return mu + eps * sd, dict(mu=mu, s2=s2)


# --------------------------------------------------------------------------- #
#  complex batch normalisation                                                #
# --------------------------------------------------------------------------- #
def _bn_axes(x):
    return (0,) + tuple(range(2, x.ndim))


def _bn_shape(x):
    return (1, x.shape[1]) + (1,) * (x.ndim - 2)


def _inv_sqrt_2x2(vuu, vuv, vvv):
    """Closed-form inverse square root of [[vuu, vuv], [vuv, vvv]],
    cplxmodule/nn/modules/batchnorm.py:108-113."""
    s = np.sqrt(vuu * vvv - vuv * vuv)
    t = s * np.sqrt(vuu + 2 * s + vvv)
    return (vvv + s) / t, -vuv / t, -vuv / t, (vuu + s) / t


def cplx_batch_norm(xr, xi, running_mean, running_var, weight=None, bias=None,
                    training=True, momentum=0.1, eps=1e-5):
    """cplx_batch_norm + whiten2x2, cplxmodule/nn/modules/batchnorm.py:62-123,
    189-278.  ``running_mean`` [2,F] / ``running_var`` [2,2,F] are updated IN
    PLACE when training (biased covariance, nugget on the diagonal only).
    Returns (yr, yi)."""
    ax, shp = _bn_axes(xr), _bn_shape(xr)
    if training or running_mean is None:
        mu, mv = xr.mean(ax), xi.mean(ax)
        if running_mean is not None:
            running_mean += momentum * (np.stack([mu, mv]) - running_mean)
    else:
        mu, mv = running_mean
    cu, cv = xr - mu.reshape(shp), xi - mv.reshape(shp)
    if training or running_var is None:
        vuu = (cu * cu).mean(ax) + np.asarray(eps, xr.dtype)
        vvv = (cv * cv).mean(ax) + np.asarray(eps, xr.dtype)
        vuv = (cu * cv).mean(ax)
        if running_var is not None:
            cov = np.stack([vuu, vuv, vuv, vvv]).reshape(2, 2, -1)
            running_var += momentum * (cov - running_var)
    else:
        vuu, vuv, _, vvv = running_var.reshape(4, -1)
    p, q, r, w = _inv_sqrt_2x2(vuu, vuv, vvv)
    zu = cu * p.reshape(shp) + cv * r.reshape(shp)
    zv = cu * q.reshape(shp) + cv * w.reshape(shp)
    if weight is not None:
        W = weight.reshape(2, 2, *shp)
        zu, zv = (zu * W[0, 0] + zv * W[0, 1] + bias[0].reshape(shp),
                  zu * W[1, 0] + zv * W[1, 1] + bias[1].reshape(shp))
    return zu, zv


def cplx_batch_norm_bwd(gr, gi, xr, xi, running_mean, running_var, weight=None,
                        training=True, eps=1e-5):
    """Gradient of ``cplx_batch_norm`` wrt (xr, xi, weight, bias), derived by
    hand through the closed-form 2x2 inverse square root (gradients DO flow
    through s and t, cf. the comment at batchnorm.py:105-107).
    ``running_*`` are the statistics used in eval mode (ignored if training)."""
    ax, shp = _bn_axes(xr), _bn_shape(xr)
    N = xr.size // xr.shape[1]
    if training:
        mu, mv = xr.mean(ax), xi.mean(ax)
    else:
        mu, mv = running_mean
    cu, cv = xr - mu.reshape(shp), xi - mv.reshape(shp)
    if training:
        a = (cu * cu).mean(ax) + np.asarray(eps, xr.dtype)
        d = (cv * cv).mean(ax) + np.asarray(eps, xr.dtype)
        b = (cu * cv).mean(ax)
    else:
        a, b, _, d = running_var.reshape(4, -1)
    p, q, r, w = _inv_sqrt_2x2(a, b, d)
    out = {}
    if weight is not None:
        zu = cu * p.reshape(shp) + cv * r.reshape(shp)
        zv = cu * q.reshape(shp) + cv * w.reshape(shp)
        out["dweight"] = np.stack([(gr * zu).sum(ax), (gr * zv).sum(ax),
                                   (gi * zu).sum(ax), (gi * zv).sum(ax)]
                                  ).reshape(2, 2, -1)
        out["dbias"] = np.stack([gr.sum(ax), gi.sum(ax)])
        W = weight.reshape(2, 2, *shp)
        gzu = gr * W[0, 0] + gi * W[1, 0]
        gzv = gr * W[0, 1] + gi * W[1, 1]
    else:
        gzu, gzv = gr, gi
    P, Q, R, Wd = (v.reshape(shp) for v in (p, q, r, w))
    if not training:
        out["dxr"], out["dxi"] = gzu * P + gzv * Q, gzu * R + gzv * Wd
        return out
    # sums that drive the gradient wrt the whitening matrix
    gp, gr_ = (gzu * cu).sum(ax), (gzu * cv).sum(ax)
    gq, gw = (gzv * cu).sum(ax), (gzv * cv).sum(ax)
    gqr = gq + gr_
    s = np.sqrt(a * d - b * b)
    tau = a + d + 2 * s
    rt = np.sqrt(tau)
    t = s * rt
    ds = dict(a=d / (2 * s), d=a / (2 * s), b=-b / s)
    gcov = {}
    for X in "adb":
        dtau = (0.0 if X == "b" else 1.0) + 2 * ds[X]
        dt_ = ds[X] * rt + s * dtau / (2 * rt)
        dp = (((1.0 if X == "d" else 0.0) + ds[X]) * t - (d + s) * dt_) / (t * t)
        dw = (((1.0 if X == "a" else 0.0) + ds[X]) * t - (a + s) * dt_) / (t * t)
        dq = (-(1.0 if X == "b" else 0.0) * t + b * dt_) / (t * t)
        gcov[X] = gp * dp + gw * dw + gqr * dq
    gA, gD, gB = (gcov[k].reshape(shp) for k in "adb")
    sgu, sgv = gzu.sum(ax).reshape(shp), gzv.sum(ax).reshape(shp)
    out["dxr"] = (gzu * P + gzv * Q + (2 * gA / N) * cu + (gB / N) * cv
                  - (P * sgu + Q * sgv) / N)
    out["dxi"] = (gzu * R + gzv * Wd + (2 * gD / N) * cv + (gB / N) * cu
                  - (R * sgu + Wd * sgv) / N)
    return out


# --------------------------------------------------------------------------- #
#  layout converters, modReLU, complex dropout           #
# --------------------------------------------------------------------------- #
def from_interleaved_real(x):
    """cplxmodule/cplx.py:451-455 along the last dim: x[..., 2k] + i x[..., 2k+1]."""
    return x[..., 0::2].copy(), x[..., 1::2].copy()


def to_interleaved_real(re, im, flatten=True):
    """cplxmodule/cplx.py:466-470 along the last dim."""
    out = np.stack([re, im], axis=-1)
    return out.reshape(*re.shape[:-1], -1) if flatten else out


def from_concatenated_real(x):
    """cplxmodule/cplx.py:458-461."""
    d = x.shape[-1] // 2
    return x[..., :d].copy(), x[..., d:].copy()


def modrelu(zr, zi, tau):
    """cplxmodule/cplx.py:613-615: z * relu(1 - tau / clamp(|z|, min=1e-5)); tau broadcasts."""
    dt = zr.dtype
    m = np.maximum(cplx_abs(zr, zi), np.asarray(1e-5, dt))
    s = np.maximum(np.asarray(1.0, dt) - np.asarray(tau, dt) / m, np.asarray(0.0, dt))
    return zr * s, zi * s


def modrelu_bwd(gr, gi, zr, zi, tau):
    """Hand-derived gradients of ``modrelu``: with dot = g . z,
    dz = g s + z dot tau / m^3 (active branch, |z| >= 1e-5: clamp passes the gradient at the
    boundary), dtau = -dot / m (active branch), summed down to tau's shape by the caller."""
    dt = zr.dtype
    tau = np.broadcast_to(np.asarray(tau, dt), zr.shape)
    az = cplx_abs(zr, zi)
    m = np.maximum(az, np.asarray(1e-5, dt))
    pre = np.asarray(1.0, dt) - tau / m
    active = pre > 0
    s = np.where(active, pre, 0).astype(dt)
    dot = gr * zr + gi * zi
    k = np.where(active & (az >= 1e-5), dot * tau / (m * m * m), 0).astype(dt)
    return dict(dzr=gr * s + zr * k, dzi=gi * s + zi * k,
                dtau=np.where(active, -dot / m, 0).astype(dt))


def cplx_dropout_mask(n, p, seed, offset):
    """Keep mask of the package's complex dropout (csrc/layout.hip): element e is kept iff word
    (e & 3) of Philox4x32-7(counter = (e >> 2, offset), key = seed) >= floor(p * 2^32)."""
    from .philox import philox4x32
    groups = (n + 3) // 4
    words = philox4x32(np.arange(groups, dtype=np.uint64), int(offset), int(seed))
    t = p * 4294967296.0
    thresh = np.uint32(0xFFFFFFFF) if t >= 4294967295.0 else np.uint32(int(t))
    return (words.reshape(-1)[:n] >= thresh)


def _pool_out(L, k, s, p, d, ceil_mode):
    num = L + 2 * p - d * (k - 1) - 1
    o = (-(-num // s) if ceil_mode else num // s) + 1
    if ceil_mode and (o - 1) * s >= L + p:
        o -= 1
    return o


def cplx_max_pool2d(zr, zi, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """cplx.max_poolnd (cplxmodule/cplx.py:1114-1175): argmax of |z| per window (first maximum in
    row-major window order, as F.max_pool2d), both parts gathered.  Returns (yr, yi, idx)."""
    (kh, kw), (ph, pw), (dh, dw) = _pair(kernel_size), _pair(padding), _pair(dilation)
    sh, sw = (kh, kw) if stride is None else _pair(stride)
    B, C, H, W = zr.shape
    Ho, Wo = _pool_out(H, kh, sh, ph, dh, ceil_mode), _pool_out(W, kw, sw, pw, dw, ceil_mode)
    mod = cplx_abs(zr, zi)
    yr, yi = np.zeros((B, C, Ho, Wo), zr.dtype), np.zeros((B, C, Ho, Wo), zr.dtype)
    idx = np.zeros((B, C, Ho, Wo), np.int64)
    for oh in range(Ho):
        for ow in range(Wo):
            best = np.full((B, C), -np.inf)
            sel = np.full((B, C), -1, np.int64)
            for i in range(kh):
                h = oh * sh - ph + i * dh
                if not 0 <= h < H:
                    continue
                for j in range(kw):
                    w = ow * sw - pw + j * dw
                    if not 0 <= w < W:
                        continue
                    m = mod[:, :, h, w]
                    take = (sel < 0) | (m > best)
                    best = np.where(take, m, best)
                    sel = np.where(take, h * W + w, sel)
            idx[:, :, oh, ow] = sel
    flat_r, flat_i = zr.reshape(B, C, -1), zi.reshape(B, C, -1)
    yr = np.take_along_axis(flat_r, idx.reshape(B, C, -1), -1).reshape(B, C, Ho, Wo)
    yi = np.take_along_axis(flat_i, idx.reshape(B, C, -1), -1).reshape(B, C, Ho, Wo)
    return yr, yi, idx


def cplx_max_pool2d_bwd(gr, gi, idx, in_shape):
    """Scatter-add of the output gradients to the selected positions."""
    B, C, H, W = in_shape
    dzr, dzi = np.zeros((B, C, H * W), gr.dtype), np.zeros((B, C, H * W), gr.dtype)
    fi = idx.reshape(B, C, -1)
    for b in range(B):
        for c in range(C):
            np.add.at(dzr[b, c], fi[b, c], gr[b, c].reshape(-1))
            np.add.at(dzi[b, c], fi[b, c], gi[b, c].reshape(-1))
    return dzr.reshape(in_shape), dzi.reshape(in_shape)
