"""Kernel wrappers (tensor in, tensor out) and the autograd Functions built on them.

Everything here launches libcplxamd.so kernels on the current HIP stream; torch is used only
to allocate outputs and to hook the kernels into autograd.  Shapes follow the reference:
complex tensors are (real, imag) pairs of equal-shaped planes (cplxmodule/cplx.py:10-52).
"""
import os

import functools

import torch

from . import _lib, x3
from ._lib import call, dtype_code, launch_flags, ptr, require_device, scratch_key, stream_ptr, try_call

_ws_cache = {}


def once_differentiable(fn):
    """Decorator of a `backward` that raw kernels compute (not differentiable in turn).  torch's own once_differentiable
    raises on a second differentiation only when an INCOMING gradient requires grad; with a constant upstream gradient
    (a penalty on d sum(y) / dx) the result would come back without history and a gradient penalty built on it would
    contribute nothing -- silently.  This one routes the results through the error node whenever the backward runs with
    create_graph=True: differentiating them again always raises (never a silently wrong second derivative)."""
    @functools.wraps(fn)
    def wrapper(ctx, *args):
        with torch.no_grad():
            outputs = fn(ctx, *args)
        if not torch.is_grad_enabled():
            return outputs
        single = not isinstance(outputs, tuple)
        outs = (outputs,) if single else outputs
        live = [i for i, v in enumerate(outs) if isinstance(v, torch.Tensor) and v.is_floating_point()]
        if not live:
            return outputs
        err = torch._C._functions.DelayedError(
            b"trying to differentiate twice a function that was marked with @once_differentiable (cplxmodule_amd: this "
            b"backward is computed by raw HIP kernels; second derivatives exist for the linear layers, Cplx products and "
            b"matmul only)", len(live))
        wrapped = err(*[outs[i].detach().requires_grad_(True) for i in live])
        wrapped = (wrapped,) if isinstance(wrapped, torch.Tensor) else wrapped
        res = list(outs)
        for i, w in zip(live, wrapped):
            res[i] = w
        return res[0] if single else tuple(res)
    return wrapper


def _ws(device):
    key = scratch_key(device)
    if key not in _ws_cache:
        nbytes = int(_lib.load().cplxamd_vd_kl_ws_bytes())
        _ws_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _ws_cache[key]


def _c(t):
    return None if t is None else t.contiguous()


def _layout_of(t):
    """The dense layout the elementwise kernels should keep for `t`: torch.channels_last when a 4-d tensor is stored
    so (and is not plain contiguous as well), else torch.contiguous_format.  The kernels walk the storage linearly, so
    any ONE dense layout shared by all operands and outputs is as good as another -- and converting a channels-last
    activation to NCHW here would break a conv -> batch-norm -> activation -> conv chain that needs no copy at all."""
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
        return torch.channels_last
    return torch.contiguous_format


def _cf(t, fmt):
    return None if t is None else t.contiguous(memory_format=fmt)


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise _lib.CplxAmdError(f"expected a float32 tensor, got {t.dtype}")
    return t


# ------------------------------------------------------------------------------------------ #
#  raw kernels                                                                               #
# ------------------------------------------------------------------------------------------ #
def cast(t, dtype):
    """dtype conversion kernel (float32 <-> bfloat16); returns `t` itself if nothing to do."""
    if t.dtype == dtype:
        return t
    require_device(t)
    t = _cf(t, _layout_of(t))             # (elementwise: any one dense layout; a channels-last tensor stays so)
    out = torch.empty_like(t, dtype=dtype)
    call("cplxamd_cast", ptr(t), ptr(out), t.numel(), dtype_code(t), dtype_code(out), stream_ptr())
    return out


def transpose2d(t):
    """[R, C] -> contiguous [C, R] through the LDS-tiled transpose kernel."""
    require_device(t)
    t = t.contiguous()
    R, C = t.shape
    out = torch.empty(C, R, dtype=t.dtype, device=t.device)
    call("cplxamd_transpose", ptr(t), C, ptr(out), R, R, C, dtype_code(t), stream_ptr())
    return out


def colsum(t, out=None):
    """sum over dim 0 of a [R, C] matrix -> float32 [C] (written into `out` when given)."""
    require_device(t)
    t = t.contiguous()
    R, C = t.shape
    if out is None or not (out.is_contiguous() and out.dtype == torch.float32 and out.numel() == C):
        dst, out = out, torch.empty(C, dtype=torch.float32, device=t.device)
    else:
        dst = None
    ws = torch.empty(int(_lib.load().cplxamd_colsum_ws_bytes(C)), dtype=torch.uint8, device=t.device)
    call("cplxamd_colsum", ptr(t), C, ptr(out), R, C, dtype_code(t), ptr(ws), stream_ptr())
    return out if dst is None else dst.copy_(out)


def attach_colsum(t, sums):
    """Remember on the tensor object that `sums` (float32 [C]) holds its per-channel sums over every other dimension:
    a producer that had all values in registers anyway (the batch-norm backward) leaves them for the consumer's bias
    gradient.  Validated on use against the storage address and the version counter."""
    t._cplxamd_colsum = (sums, t.data_ptr(), t._version, tuple(t.shape))


def colsum_hint(t):
    """The sums attach_colsum left on exactly this tensor (same storage, unmodified since), or None."""
    h = getattr(t, "_cplxamd_colsum", None)
    if h is not None and h[1] == t.data_ptr() and h[2] == t._version and h[3] == tuple(t.shape) and h[0].numel() == t.shape[1]:
        return h[0]
    return None


def attach_scale(tr, ti, scale):
    """Remember on the two float32 planes of a complex tensor that `scale` (device float32[2] = {s, 1 / s}, x3.scale_of's
    format) is the power-of-two scale of max |.| over both: the producer (the batch-norm backward's apply pass) had every
    value in registers; the consumer that cuts the planes into half pieces (x3 'x2') skips its absmax pass.  Validated on
    use like attach_colsum."""
    tr._cplxamd_scale = (scale, tr.data_ptr(), tr._version, ti.data_ptr(), ti._version, tuple(tr.shape))


def scale_hint(tr, ti):
    """The scale attach_scale left for exactly this pair of planes (same storage, unmodified since), or None."""
    h = getattr(tr, "_cplxamd_scale", None)
    if (h is not None and h[1] == tr.data_ptr() and h[2] == tr._version and h[3] == ti.data_ptr() and h[4] == ti._version
            and h[5] == tuple(tr.shape) == tuple(ti.shape)):
        return h[0]
    return None


def attach_wgrad(tr, ti, dw):
    """Remember on the two planes of a convolution's output gradient that `dw` = (dW_r, dW_i) is the weight gradient of
    exactly these planes against the input of the convolution whose autograd node is `node` (bn.py: the batch-norm
    backward that formed the planes inside the weight-gradient kernel).  Validated on use like attach_colsum."""
    tr._cplxamd_wgrad = (dw, tr.data_ptr(), tr._version, ti.data_ptr(), ti._version, tuple(tr.shape))


def wgrad_hint(tr, ti, node):
    """(dW_r, dW_i) attach_wgrad left for exactly this pair of planes and this convolution node, or None."""
    h = getattr(tr, "_cplxamd_wgrad", None)
    if (h is not None and h[1] == tr.data_ptr() and h[2] == tr._version and h[3] == ti.data_ptr() and h[4] == ti._version
            and h[5] == tuple(tr.shape) == tuple(ti.shape) and h[0][2] is node):
        return h[0][0], h[0][1]
    return None


def attach_moments(tr, ti, partials, chunks):
    """Remember on the two planes of a convolution output that `partials` ([chunks][C][5] float64) holds the batch-norm
    forward moments of exactly these tensors (conv.cl_conv, csrc/conv_cl2.hip MOM epilogue): the batch-norm layer that
    consumes them skips its own moment pass.  Validated on use like attach_colsum."""
    tr._cplxamd_moments = (partials, int(chunks), tr.data_ptr(), tr._version, ti.data_ptr(), ti._version, tuple(tr.shape))


def moments_hint(tr, ti):
    """(partials, chunks) attach_moments left for exactly this pair of planes (same storage, unmodified since), or None."""
    h = getattr(tr, "_cplxamd_moments", None)
    if (h is not None and h[2] == tr.data_ptr() and h[3] == tr._version and h[4] == ti.data_ptr() and h[5] == ti._version
            and h[6] == tuple(tr.shape) == tuple(ti.shape)):
        return h[0], h[1]
    return None


def colsum2(tr, ti, out=None):
    """Column sums of two planes (the complex bias gradient) -> float32 ([C], [C])."""
    if out is None and tr.dim() == 2 and tr.shape == ti.shape and tr.dtype == ti.dtype:
        require_device(tr, ti)
        tr, ti = tr.contiguous(), ti.contiguous()
        R, C = tr.shape
        o = torch.empty(2, C, dtype=torch.float32, device=tr.device)
        ws = torch.empty(int(_lib.load().cplxamd_colsum_ws_bytes(C)), dtype=torch.uint8, device=tr.device)
        call("cplxamd_colsum2", ptr(tr), ptr(ti), C, ptr(o[0]), ptr(o[1]), R, C, dtype_code(tr), ptr(ws), stream_ptr())
        return o[0], o[1]
    return colsum(tr, None if out is None else out[0]), colsum(ti, None if out is None else out[1])


def abs2(xr, xi=None, out_dtype=None):
    """xr^2 + xi^2 (or xr^2), cplxmodule/nn/relevance/complex/base.py:51."""
    require_device(xr, xi)
    fmt = _layout_of(xr)
    xr, xi = _cf(xr, fmt), _cf(xi, fmt)
    out = torch.empty_like(xr, dtype=out_dtype or xr.dtype)
    call("cplxamd_abs2", ptr(xr), ptr(xi), ptr(out), xr.numel(), dtype_code(xr),
         dtype_code(out), stream_ptr())
    return out


def modulus(xr, xi):
    """abs(Cplx), cplxmodule/cplx.py:183-192 (float32 or bfloat16 planes)."""
    require_device(xr, xi)
    fmt = _layout_of(xr)
    xr, xi = _cf(xr, fmt), _cf(xi, fmt)
    out = torch.empty_like(xr)
    call("cplxamd_cplx_abs_fwd", ptr(xr), ptr(xi), ptr(out), xr.numel(), dtype_code(xr), stream_ptr())
    return out


def mask_mul(tr, ti, mask, out_dtype=None):
    """(tr * mask, ti * mask) in one pass, converted to `out_dtype`; ti may be None (real layers)."""
    require_device(tr, ti, mask)
    tr, ti, mask = _c(tr), _c(ti), _f32(_c(mask.expand_as(tr)))
    odt = out_dtype or tr.dtype
    our = torch.empty_like(tr, dtype=odt)
    oui = None if ti is None else torch.empty_like(ti, dtype=odt)
    call("cplxamd_mask_mul", ptr(tr), ptr(ti), ptr(mask), ptr(our), ptr(oui), tr.numel(), dtype_code(tr),
         dtype_code(our), stream_ptr())
    return our, oui


def exp(x, out_dtype=torch.float32):
    require_device(x)
    x = _f32(_c(x))
    out = torch.empty_like(x, dtype=out_dtype)
    call("cplxamd_exp", ptr(x), ptr(out), x.numel(), dtype_code(out), stream_ptr())
    return out


_gemm_ws_cache = {}


def _gemm_ws(M, N, K, cplx, a, c):
    """Split-K scratch (asked for by few-tile / long-K GEMMs: wgrad at large batch, small heads)."""
    # (half pieces run the bf16 path's kernels compiled for the half MFMA: same split-K plan)
    need = int(_lib.load().cplxamd_gemm_ws_bytes(M, N, K, int(cplx), _lib.BF16 if a.dtype == torch.float16 else dtype_code(a),
                                                 dtype_code(c)))
    if need == 0:
        return None
    key = scratch_key(a.device)
    buf = _gemm_ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = _gemm_ws_cache[key] = torch.empty(need, dtype=torch.uint8, device=a.device)
    return buf


_gauss_ws_cache = {}


def _gauss_ws(M, N, K, device):
    need = int(_lib.load().cplxamd_cgemm3m_ws_bytes(M, N, K))
    key = scratch_key(device)
    buf = _gauss_ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = _gauss_ws_cache[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return buf


def gauss_ok(M, N, K):
    """Shapes the 3M entry accepts (mirrors launch_gemm_bf16_gauss); the layer-level helpers run
    4M for the rest, the C entry point itself refuses them."""
    return K >= 32 and K % 32 == 0 and N % 4 == 0 and (M * K) % 8 == 0 and (N * K) % 8 == 0


def _beta(beta):
    """Device scalar for the scaled accumulate (None: plain +=)."""
    if beta is None:
        return None
    require_device(beta)
    return _f32(beta.reshape(()).contiguous()) if beta.dtype == torch.float32 else beta.reshape(()).float()


def cgemm(ar, ai, a_strides, br, bi, b_strides, M, N, K, bias=None, conj_b=False,
          out_dtype=torch.float32, out=None, accumulate=False, algo=0, beta=None, emul=None, scales=None):
    """C[m,n] = sum_k A[m,k] op(B[n,k]) (+ bias[n]) on planar complex operands.
    `a_strides` / `b_strides` are (row, col) element strides into the given planes.
    algo: 0 = 4M (one fused K loop), 1 = Gauss 3M (dense bf16 operands only).
    accumulate: C = result + beta * C with `beta` a 0-d DEVICE tensor (None: 1);
    emul: float32 [M,N] multiplier of both result planes (float32 output only)."""
    require_device(ar, ai, br, bi, emul)
    if out is None:
        cr = torch.empty(M, N, dtype=out_dtype, device=ar.device)
        ci = torch.empty(M, N, dtype=out_dtype, device=ar.device)
    else:
        cr, ci = out
    b_r, b_i = (None, None) if bias is None else bias
    ws = _gauss_ws(M, N, K, ar.device) if algo == 1 else _gemm_ws(M, N, K, True, ar, cr)
    beta = _beta(beta) if accumulate else None
    if ar.dtype == torch.float16:
        # IEEE-half pieces of the float32 split products (x3.py 'x2'): float32 out, `scales` = the operands' device
        # {s, 1 / s} pairs, undone behind the K loop
        sa, sb = scales if scales is not None else (None, None)
        call("cplxamd_cgemm_sc_fl", ptr(ar), ptr(ai), a_strides[0], a_strides[1], ptr(br), ptr(bi), b_strides[0], b_strides[1],
             ptr(b_r), ptr(b_i), ptr(emul), ptr(cr), ptr(ci), N, M, N, K, int(conj_b), _lib.F16, int(accumulate), ptr(beta),
             ptr(sa), ptr(sb), ptr(ws), 0 if ws is None else ws.numel(), launch_flags(), stream_ptr())
        return cr, ci
    call("cplxamd_cgemm_fl", ptr(ar), ptr(ai), a_strides[0], a_strides[1], ptr(br), ptr(bi),
         b_strides[0], b_strides[1], ptr(b_r), ptr(b_i), ptr(emul), ptr(cr), ptr(ci), N, M, N, K,
         int(conj_b), dtype_code(ar), dtype_code(cr), int(accumulate), ptr(beta), int(algo), ptr(ws),
         0 if ws is None else ws.numel(), launch_flags(), stream_ptr())
    return cr, ci


def cgemm_batched(ar, ai, a_strides, br, bi, b_strides, batch, M, N, K, conj_b=False, out_dtype=None):
    """`batch` complex products A[z] B[z]^T in one launch; *_strides = (row, col, batch) in elements."""
    require_device(ar, ai, br, bi)
    odt = out_dtype or ar.dtype
    cr = torch.empty(batch, M, N, dtype=odt, device=ar.device)
    ci = torch.empty_like(cr)
    call("cplxamd_cgemm_batched", ptr(ar), ptr(ai), *a_strides, ptr(br), ptr(bi), *b_strides, ptr(cr),
         ptr(ci), N, M * N, batch, M, N, K, int(conj_b), dtype_code(ar), dtype_code(cr), stream_ptr())
    return cr, ci


def rgemm(a, a_strides, b, b_strides, M, N, K, bias=None, emul=None, out_dtype=torch.float32,
          out=None, emul_exp=False, accumulate=False, beta=None, scales=None):
    """C = (A B^T + bias) * emul (emul_exp: * exp(emul)); accumulate: C = that + beta * C."""
    require_device(a, b, bias, emul)
    c = torch.empty(M, N, dtype=out_dtype, device=a.device) if out is None else out
    ws = _gemm_ws(M, N, K, False, a, c)
    beta = _beta(beta) if accumulate else None
    if a.dtype == torch.float16:              # (see cgemm)
        sa, sb = scales if scales is not None else (None, None)
        call("cplxamd_rgemm_sc_fl", ptr(a), a_strides[0], a_strides[1], ptr(b), b_strides[0], b_strides[1], ptr(bias), ptr(emul),
             int(emul_exp), ptr(c), N, M, N, K, _lib.F16, int(accumulate), ptr(beta), ptr(sa), ptr(sb), ptr(ws),
             0 if ws is None else ws.numel(), launch_flags(), stream_ptr())
        return c
    call("cplxamd_rgemm_fl", ptr(a), a_strides[0], a_strides[1], ptr(b), b_strides[0], b_strides[1],
         ptr(bias), ptr(emul), int(emul_exp), ptr(c), N, M, N, K, dtype_code(a), dtype_code(c),
         int(accumulate), ptr(beta), ptr(ws), 0 if ws is None else ws.numel(), launch_flags(), stream_ptr())
    return c


def philox_normal(n, seed, offset, device, complex_=False):
    """The in-kernel Philox noise stream, materialised (tests / debugging)."""
    er = torch.empty(n, dtype=torch.float32, device=device)
    ei = torch.empty(n, dtype=torch.float32, device=device) if complex_ else None
    require_device(er)
    call("cplxamd_philox_normal", ptr(er), ptr(ei), seed, offset, n, stream_ptr())
    return (er, ei) if complex_ else er


def _noise_args(seed, offset):
    """(seed, offset, device-state pointer): `seed` may be a device int64[2] tensor
    {seed, offset} (graph-capturable noise position) instead of a host integer."""
    if isinstance(seed, torch.Tensor):
        return 0, 0, ptr(seed)
    return seed, offset, None


def philox_advance(state):
    """Copy the device noise position and advance it by one stochastic forward."""
    used = torch.empty_like(state)
    call("cplxamd_philox_advance", ptr(state), ptr(used), stream_ptr())
    return used


def _al16(t):
    """The streaming kernels move 16 bytes per lane: re-home the rare view that is not aligned."""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone()


def _s2(s2, like):
    """The variance operand of the noise injection: float32, or bf16 as the variance GEMM / convolution of a bf16
    layer wrote it (kept as it is: no float32 copy of a [B, O] tensor)."""
    return s2 if (s2.dtype == torch.bfloat16 and like.dtype == torch.bfloat16) else _f32(s2)


def _s2_dtype(x):
    return torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32


def reparam_fwd(mu_r, mu_i, s2, eps=None, seed=0, offset=0, inplace=False):
    """y = mu + eps * sqrt(max(s2, 1e-8)); eps=(eps_r, eps_i) / eps_r tensor or None (Philox)."""
    require_device(mu_r, mu_i, s2)
    fmt = _layout_of(mu_r)               # all operands in the layout of mu (the kernel walks the storage linearly)
    mu_r, mu_i, s2 = _al16(_cf(mu_r, fmt)), _al16(_cf(mu_i, fmt)), _al16(_s2(_cf(s2, fmt), mu_r))
    e_r = e_i = None
    if eps is not None:
        e_r, e_i = eps if isinstance(eps, (tuple, list)) else (eps, None)
        e_r, e_i = _cf(cast(e_r, mu_r.dtype), fmt), (None if e_i is None else _cf(cast(e_i, mu_r.dtype), fmt))
        e_r, e_i = _al16(e_r), _al16(e_i)
    y_r = mu_r if inplace else torch.empty_like(mu_r)
    y_i = None if mu_i is None else (mu_i if inplace else torch.empty_like(mu_i))
    sd, of, st = _noise_args(seed, offset)
    call("cplxamd_lrt_reparam_fwd_ex", ptr(mu_r), ptr(mu_i), ptr(s2), ptr(e_r), ptr(e_i), sd,
         of, st, ptr(y_r), ptr(y_i), mu_r.numel(), dtype_code(mu_r), dtype_code(s2), stream_ptr())
    return y_r, y_i


def _colsum_ws(device, nbytes):
    key = scratch_key(device)
    buf = _colsum_ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _colsum_ws_cache[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return buf


_colsum_ws_cache = {}


def _as_rows(t, rows, cols):
    """[rows, cols] view of the storage of a dense tensor ([B, O] as it is, channels-last planes as [B H W][C])."""
    if t.dim() == 4 and not t.is_contiguous():
        t = t.permute(0, 2, 3, 1)
    return t.reshape(rows, cols)


def reparam_bwd(g_r, g_i, s2, eps=None, seed=0, offset=0, out_dtype=torch.float32, bias_sums=None):
    """d s2 of the noise injection.  bias_sums = (rows, cols[, (out_r, out_i)]): the operands are [rows][cols] matrices as
    stored (a [B, O] gradient; channels-last planes as [B H W][C]) and the per-column sums of g_r / g_i -- the bias
    gradient of the layer -- come out of the same pass: returns (g_s2, sum_r, sum_i); shapes the fused kernel does not
    take run the flat kernel + the column-sum kernels."""
    require_device(g_r, g_i, s2)
    fmt = _layout_of(s2)                 # the layout the forward ran in (s2 is its saved tensor)
    g_r, g_i, s2 = _al16(_cf(g_r, fmt)), _al16(_cf(g_i, fmt)), _al16(_s2(_cf(s2, fmt), g_r))
    e_r = e_i = None
    if eps is not None:
        e_r, e_i = eps if isinstance(eps, (tuple, list)) else (eps, None)
        e_r, e_i = _cf(cast(e_r, g_r.dtype), fmt), (None if e_i is None else _cf(cast(e_i, g_r.dtype), fmt))
        e_r, e_i = _al16(e_r), _al16(e_i)
    g_s2 = torch.empty_like(s2, dtype=out_dtype)
    sd, of, st = _noise_args(seed, offset)
    if bias_sums is not None:
        rows, cols = int(bias_sums[0]), int(bias_sums[1])
        assert rows * cols == g_r.numel()
        outs = bias_sums[2] if len(bias_sums) > 2 and bias_sums[2] is not None else (None, None)
        ok = lambda t: t is not None and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == cols  # noqa: E731
        sum_r = outs[0] if ok(outs[0]) else torch.empty(cols, dtype=torch.float32, device=g_r.device)
        sum_i = None if g_i is None else (outs[1] if ok(outs[1]) else torch.empty(cols, dtype=torch.float32, device=g_r.device))
        need = int(_lib.load().cplxamd_lrt_reparam_bwd_cols_ws_bytes(rows, cols))
        fused = need > 0
        if fused:
            ws = _colsum_ws(g_r.device, need)
            fused = try_call("cplxamd_lrt_reparam_bwd_cols", ptr(g_r), ptr(g_i), ptr(s2), ptr(e_r), ptr(e_i), sd, of, st,
                             ptr(g_s2), rows, cols, dtype_code(g_r), dtype_code(g_s2), dtype_code(s2), ptr(sum_r),
                             ptr(sum_i), ptr(ws), ws.numel(), stream_ptr())
        if not fused:
            call("cplxamd_lrt_reparam_bwd_ex", ptr(g_r), ptr(g_i), ptr(s2), ptr(e_r), ptr(e_i), sd, of, st,
                 ptr(g_s2), g_r.numel(), dtype_code(g_r), dtype_code(g_s2), dtype_code(s2), stream_ptr())
            if g_i is not None:                  # both planes' bias sums in one launch
                a_r, a_i = _as_rows(g_r, rows, cols).contiguous(), _as_rows(g_i, rows, cols).contiguous()
                cws = torch.empty(int(_lib.load().cplxamd_colsum_ws_bytes(cols)), dtype=torch.uint8, device=g_r.device)
                call("cplxamd_colsum2", ptr(a_r), ptr(a_i), cols, ptr(sum_r), ptr(sum_i), rows, cols, dtype_code(a_r), ptr(cws),
                     stream_ptr())
            else:
                colsum(_as_rows(g_r, rows, cols), out=sum_r)
        for dst, src in zip(outs, (sum_r, sum_i)):
            if dst is not None and src is not None and dst is not src:
                dst.copy_(src.view_as(dst))
        return g_s2, (outs[0] if outs[0] is not None else sum_r), (None if g_i is None else (outs[1] if outs[1] is not None else sum_i))
    call("cplxamd_lrt_reparam_bwd_ex", ptr(g_r), ptr(g_i), ptr(s2), ptr(e_r), ptr(e_i), sd, of, st,
         ptr(g_s2), g_r.numel(), dtype_code(g_r), dtype_code(g_s2), dtype_code(s2), stream_ptr())
    return g_s2


def lrt_dx_accum(dxr, dxi, xr, xi, ga):
    require_device(dxr, dxi, xr, xi, ga)
    call("cplxamd_lrt_dx_accum", ptr(dxr), ptr(dxi), ptr(xr), ptr(xi), ptr(ga), dxr.numel(),
         dtype_code(dxr), dtype_code(ga), stream_ptr())


def kl_fwd(kind, wr, wi, ls2, elementwise=False, total=True):
    require_device(wr, wi, ls2)
    wr, wi, ls2 = _f32(_c(wr)), _f32(_c(wi)), _f32(_c(ls2))
    elem = torch.empty_like(wr) if elementwise else None
    tot = torch.empty((), dtype=torch.float32, device=wr.device) if total else None
    call("cplxamd_vd_kl_fwd", ptr(wr), ptr(wi), ptr(ls2), _lib.KL_KINDS[kind], ptr(elem), ptr(tot),
         ptr(_ws(wr.device)), wr.numel(), stream_ptr())
    return elem, tot


def kl_bwd(kind, wr, wi, ls2, g_elem=None, g_scalar=None, need=(True, True, True)):
    require_device(wr, wi, ls2, g_elem, g_scalar)
    wr, wi, ls2 = _f32(_c(wr)), _f32(_c(wi)), _f32(_c(ls2))
    g_elem = _f32(_c(g_elem))
    if g_scalar is not None:
        g_scalar = _f32(g_scalar.reshape(()).contiguous())
    g_ls2 = torch.empty_like(ls2) if need[0] else None
    g_wr = torch.empty_like(wr) if need[1] else None
    g_wi = torch.empty_like(wi) if (wi is not None and need[2]) else None
    call("cplxamd_vd_kl_bwd", ptr(wr), ptr(wi), ptr(ls2), _lib.KL_KINDS[kind], ptr(g_elem),
         ptr(g_scalar), ptr(g_ls2), ptr(g_wr), ptr(g_wi), wr.numel(), stream_ptr())
    return g_ls2, g_wr, g_wi


def kl_fwd_bwd(kind, wr, wi, ls2, gscale=1.0):
    """One pass: (sum(penalty), gscale * d sum / d(log_sigma2, wr, wi))."""
    require_device(wr, wi, ls2)
    wr, wi, ls2 = _f32(_c(wr)), _f32(_c(wi)), _f32(_c(ls2))
    tot = torch.empty((), dtype=torch.float32, device=wr.device)
    g_ls2, g_wr = torch.empty_like(ls2), torch.empty_like(wr)
    g_wi = None if wi is None else torch.empty_like(wi)
    call("cplxamd_vd_kl_fwd_bwd", ptr(wr), ptr(wi), ptr(ls2), _lib.KL_KINDS[kind], float(gscale),
         ptr(tot), ptr(g_ls2), ptr(g_wr), ptr(g_wi), ptr(_ws(wr.device)), wr.numel(), stream_ptr())
    return tot, g_ls2, g_wr, g_wi


def prep_kl(kind, wr, wi, ls2, with_kl, grads=None):
    """ONE pass over the float32 parameters of a bf16 VD / ARD layer: bf16 weight planes, bf16
    exp(log_sigma2) and -- with_kl -- sum(penalty) with its unscaled gradients written to
    `grads` = (g_ls2, g_wr, g_wi) (allocated here when None).  wi None: real layer.
    -> (wr_bf16, wi_bf16, S_bf16, kl_total or None, grads or None)"""
    require_device(wr, wi, ls2)
    bf = torch.bfloat16
    wb = torch.empty_like(wr, dtype=bf)
    wib = None if wi is None else torch.empty_like(wi, dtype=bf)
    sb = torch.empty_like(ls2, dtype=bf)
    tot = None
    if with_kl:
        tot = torch.empty((), dtype=torch.float32, device=wr.device)
        if grads is None:
            grads = (torch.empty_like(ls2), torch.empty_like(wr), None if wi is None else torch.empty_like(wi))
    g = grads if with_kl else (None, None, None)
    call("cplxamd_vd_prep_kl", ptr(wr), ptr(wi), ptr(ls2), _lib.KL_KINDS.get(kind, 0), int(bool(with_kl)),
         ptr(wb), ptr(wib), ptr(sb), ptr(tot), ptr(g[0]), ptr(g[1]), ptr(g[2]),
         ptr(_ws(wr.device)) if with_kl else None, wr.numel(), stream_ptr())
    return wb, wib, sb, tot, (grads if with_kl else None)


def _prep_ok(x, *params):
    """The fused preparation kernel takes dense float32 parameters, element count % 4 == 0, bf16 activations."""
    return (x.dtype == torch.bfloat16 and params[0].numel() % 4 == 0 and
            all(p is None or (p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0)
                for p in params))


def log_alpha_bwd(g, wr, wi):
    require_device(g, wr, wi)
    g, wr, wi = _f32(_c(g)), _f32(_c(wr)), _f32(_c(wi))
    g_wr = torch.empty_like(wr)
    g_wi = None if wi is None else torch.empty_like(wi)
    call("cplxamd_vd_log_alpha_bwd", ptr(g), ptr(wr), ptr(wi), ptr(g_wr), ptr(g_wi), wr.numel(), stream_ptr())
    return g_wr, g_wi


def log_alpha(wr, wi, ls2):
    require_device(wr, wi, ls2)
    wr, wi, ls2 = _f32(_c(wr)), _f32(_c(wi)), _f32(_c(ls2))
    out = torch.empty_like(ls2)
    call("cplxamd_vd_log_alpha", ptr(wr), ptr(wi), ptr(ls2), ptr(out), wr.numel(), stream_ptr())
    return out


def relevance_mask(wr, wi, ls2, threshold, count=False):
    """float 0/1 mask of (log_alpha <= threshold) and, optionally, the on-device count of ones."""
    require_device(wr, wi, ls2)
    wr, wi, ls2 = _f32(_c(wr)), _f32(_c(wi)), _f32(_c(ls2))
    mask = torch.empty_like(ls2)
    cnt = torch.empty((), dtype=torch.int64, device=wr.device) if count else None
    call("cplxamd_vd_mask", ptr(wr), ptr(wi), ptr(ls2), float(threshold), ptr(mask), ptr(cnt),
         ptr(_ws(wr.device)), wr.numel(), stream_ptr())
    return (mask, cnt) if count else mask


# ------------------------------------------------------------------------------------------ #
#  linear algebra with layout handling                                                       #
# ------------------------------------------------------------------------------------------ #
def _is_bf16(t):
    return t.dtype == torch.bfloat16


class _Pieces:
    """16-bit pieces (x3.split, A side) of float32 planes, made on first use and shared by the products of one pass that
    read them (the output gradient feeds the weight gradient AND the input gradient).  `abs2`: ONE plane of pieces of
    xr^2 + xi^2 (xi None: xr^2) -- the |x|^2 operand of the variance products, never materialised in float32.
    `made`: pieces kept from the forward pass."""

    def __init__(self, *planes, abs2=False, made=None):
        self.planes, self.abs2, self.v = planes, abs2, made

    def get(self, kind):
        if self.v is None or self.v[0].kind != kind:
            if self.abs2:
                self.v = (x3.split(self.planes[0], op=x3.OP_ABS2, t2=self.planes[1] if len(self.planes) > 1 else None,
                                   kind=kind),)
            else:
                self.v = x3.split_planes(self.planes, kind=kind)
        return self.v

    def saved(self):
        """(piece tensors..., shared scale or None) for ctx.save_for_backward; (None, ..) when nothing was made."""
        if self.v is None:
            return (None,) * (len(self.planes) if not self.abs2 else 1) + (None,)
        return tuple(p.t for p in self.v) + (self.v[0].scale,)

    @staticmethod
    def restored(kind, tensors, scale):
        if kind is None or tensors[0] is None:
            return None
        n = 3 if kind == "x3" else 2
        return tuple(x3.Pieces(t, kind, scale, n) for t in tensors)


# float32 split products: keep the input's bf16 pieces (6 bytes per element and plane, + 6 for |x|^2) from the forward for
# the backward's weight gradients instead of splitting again (one HBM pass per plane less; CPLXAMD_X3_SAVE=0: remake them)
_X3_SAVE = os.environ.get("CPLXAMD_X3_SAVE", "1") != "0"


def _cplx_linear_fwd(x2r, x2i, wr, wi, bias, algo=0, mode=None, xs=None):
    """[B,I] x [O,I]^T -> [B,O]; weights are cast to the activation dtype (bf16 MFMA path).  float32 activations: split
    operands on the bf16 pipe where x3.take says so (float32-level results), else the exact float32-MFMA kernel."""
    B, I = x2r.shape
    O = wr.shape[0]
    kind = x3.take(B, O, I, x2r, x2i, wr, wi, mode=mode) if x2r.dtype == torch.float32 else None
    if kind:
        Xs = (xs or _Pieces(x2r, x2i)).get(kind)
        Ws = x3.split_planes((wr, wi), x3.SPLIT_B, kind=kind)
        yr, yi = x3.gemm_nn(Xs, Ws, B, O, I, bias=bias)
        return yr, yi, (wr, wi)
    wcr, wci = cast(wr, x2r.dtype), cast(wi, x2r.dtype)
    yr, yi = cgemm(x2r, x2i, (I, 1), wcr, wci, (I, 1), B, O, I, bias=bias, out_dtype=x2r.dtype,
                   algo=algo if gauss_ok(B, O, I) else 0)
    return yr, yi, (wcr, wci)


def _cplx_linear_dx(g2r, g2i, wr, wi, out_dtype, algo=0, mode=None, gs=None):
    """dX = G conj(W):  dX[b,i] = sum_o G[b,o] conj(W[o,i]).  The weight is read as stored
    ([O, I] = K-major for this product): no transposed copy."""
    B, O = g2r.shape
    I = wr.shape[1]
    if _is_bf16(g2r):
        wr, wi = cast(wr, torch.bfloat16), cast(wi, torch.bfloat16)
    elif out_dtype == torch.float32 and x3.take(B, I, O, g2r, g2i, wr, wi, mode=mode):
        kind = x3.take(B, I, O, mode=mode)
        Gs = (gs or _Pieces(g2r, g2i)).get(kind)
        Wst = x3.split_planes((wr, wi), x3.SPLIT_B, stacked=True, kind=kind)
        return x3.gemm_nt(Gs, Wst, B, I, O, conj_b=True)
    return cgemm(g2r, g2i, (O, 1), wr, wi, (1, I), B, I, O, conj_b=True, out_dtype=out_dtype,
                 algo=algo if gauss_ok(B, I, O) and I % 8 == 0 else 0)


_LRT_DX_FUSE = os.environ.get("CPLXAMD_LRT_DX_FUSE", "1") != "0"     # (A/B switch; results are bit-identical)
_EARLY_W = os.environ.get("CPLXAMD_DP_EARLY_W", "1") != "0"           # (A/B switch: announce dW before the variance dW)


def _cplx_lrt_dx(g2r, g2i, wr, wi, x2r, x2i, ga, mode=None, gs=None):
    """Input gradient of the complex LRT layer: dX = G conj(W) + 2 X (*) ga in ONE launch when the persistent bf16
    kernel takes the shape (cplxamd_cgemm_lrt_dx: the elementwise term rides in its epilogue), otherwise the GEMM and
    the accumulate pass -- bit-identical results either way (tests/test_gpu_r03.py)."""
    B, O = g2r.shape
    I = wr.shape[1]
    if (_LRT_DX_FUSE and _is_bf16(g2r) and _is_bf16(x2r) and _is_bf16(ga) and _is_bf16(wr) and I % 8 == 0 and O % 8 == 0
            and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (g2r, g2i, wr, wi, x2r, x2i, ga))):
        dxr = torch.empty(B, I, dtype=torch.bfloat16, device=g2r.device)
        dxi = torch.empty_like(dxr)
        if try_call("cplxamd_cgemm_lrt_dx_fl", ptr(g2r), ptr(g2i), O, 1, ptr(wr), ptr(wi), 1, I, ptr(x2r), ptr(x2i), ptr(ga), I,
                    ptr(dxr), ptr(dxi), I, B, I, O, dtype_code(g2r), launch_flags(), stream_ptr()):
            return dxr, dxi
    dxr, dxi = _cplx_linear_dx(g2r, g2i, wr, wi, x2r.dtype, mode=mode, gs=gs)
    lrt_dx_accum(dxr, dxi, x2r, x2i, ga)
    return dxr, dxi


def _real_lrt_dx(g2, w, x2, ga, mode=None, gs=None):
    """Input gradient of the real LRT layer, dX = G W + 2 X (*) ga: one launch (cplxamd_rgemm_lrt_dx) when the persistent
    bf16 kernel takes the shape, else the GEMM and the accumulate pass (bit-identical)."""
    B, O = g2.shape
    I = w.shape[1]
    if (_LRT_DX_FUSE and _is_bf16(g2) and _is_bf16(x2) and _is_bf16(ga) and _is_bf16(w) and I % 8 == 0 and O % 8 == 0
            and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (g2, w, x2, ga))):
        dx = torch.empty(B, I, dtype=torch.bfloat16, device=g2.device)
        if try_call("cplxamd_rgemm_lrt_dx_fl", ptr(g2), O, 1, ptr(w), 1, I, ptr(x2), ptr(ga), I, ptr(dx), I, B, I, O,
                    dtype_code(g2), launch_flags(), stream_ptr()):
            return dx
    dx = _real_linear_dx(g2, w, x2.dtype, mode=mode, gs=gs)
    lrt_dx_accum(dx, None, x2, None, ga)
    return dx


def _cplx_linear_dw(g2r, g2i, x2r, x2i, out=None, algo=0, accumulate=False, beta=None, emul=None, mode=None, gs=None,
                    xs=None):
    """dW = G^T conj(X):  dW[o,i] = sum_b G[b,o] conj(X[b,i]) -> float32 [O,I]; both operands
    are K-major as stored (the bf16 kernel reads them through ds_read_b64_tr_b16).
    accumulate: out = dW + beta * out (beta a device scalar, None = 1)."""
    B, O = g2r.shape
    I = x2r.shape[1]
    kind = x3.take(O, I, B, g2r, g2i, x2r, x2i, mode=mode) if g2r.dtype == torch.float32 else None
    if kind:
        return x3.gemm_tt((gs or _Pieces(g2r, g2i)).get(kind), (xs or _Pieces(x2r, x2i)).get(kind), O, I, B, conj_b=True,
                          out=out, accumulate=accumulate, beta=beta, emul=emul)
    plain = not accumulate and emul is None
    return cgemm(g2r, g2i, (1, O), x2r, x2i, (1, I), O, I, B, conj_b=True, out=out,
                 accumulate=accumulate, beta=beta, emul=emul,
                 algo=algo if plain and gauss_ok(O, I, B) and O % 8 == 0 and I % 8 == 0 else 0)


def _real_linear_fwd(x2, w, bias, out_dtype=None, mode=None, xs=None):
    """x W^T (+ bias) for float32 or bf16 activations -> (y, the weight as the input gradient wants it)."""
    B, I = x2.shape
    O = w.shape[0]
    kind = x3.take(B, O, I, x2, w, mode=mode) if x2.dtype == torch.float32 else None
    if kind:
        return x3.gemm_nn((xs or _Pieces(x2)).get(kind), x3.split_planes((w,), x3.SPLIT_B, kind=kind), B, O, I, bias=bias), w
    wm = cast(w, x2.dtype)
    return rgemm(x2, (I, 1), wm, (I, 1), B, O, I, bias=bias, out_dtype=out_dtype or x2.dtype), wm


def _real_linear_dx(g2, w, out_dtype, mode=None, gs=None, w_exp=False):
    """G W -> [B, I].  `w_exp` (float32 split products only): the weight operand is exp(w) -- the variance path's
    sigma^2 = exp(log_sigma2), formed inside the split pass."""
    B, O = g2.shape
    I = w.shape[1]
    if _is_bf16(g2):
        w = cast(w, torch.bfloat16)
    elif out_dtype == torch.float32 and x3.take(B, I, O, g2, w, mode=mode):
        kind = x3.take(B, I, O, mode=mode)
        Wst = x3.split(w, x3.SPLIT_B, op=x3.OP_EXP if w_exp else x3.OP_ID, stacked=True, kind=kind)
        return x3.gemm_nt((gs or _Pieces(g2)).get(kind), (Wst,), B, I, O)
    if w_exp:
        w = exp(w)
    return rgemm(g2, (O, 1), w, (1, I), B, I, O, out_dtype=out_dtype)


def _real_linear_dw(g2, x2, emul=None, out=None, emul_exp=False, accumulate=False, beta=None, mode=None, gs=None, xs=None):
    """G^T X (* emul) -> float32 [O, I].  x2 may be None when `xs` (pieces of it, e.g. of |x|^2) is given and the
    split products take the shape (the caller checked with x3.take)."""
    B, O = g2.shape
    I = x2.shape[1] if x2 is not None else xs.planes[0].shape[1]
    kind = x3.take(O, I, B, g2, x2, mode=mode) if g2.dtype == torch.float32 else None
    if kind:
        return x3.gemm_tt((gs or _Pieces(g2)).get(kind), (xs or _Pieces(x2)).get(kind), O, I, B, out=out,
                          accumulate=accumulate, beta=beta, emul=emul, emul_exp=emul_exp)
    if g2.dtype != x2.dtype:
        g2, x2 = cast(g2, torch.float32), cast(x2, torch.float32)
    return rgemm(g2, (1, O), x2, (1, I), O, I, B, emul=emul, out=out, emul_exp=emul_exp,
                 accumulate=accumulate, beta=beta)


def _scaled(beta, t):
    return None if t is None else t * beta


def _saved(ctx):
    """ctx.saved_tensors of an LRT node whose KL output shares the node (KLFusion).  A backward pass of the KL term
    alone goes through this node too, and autograd then frees the node's saved activations like after any backward
    pass: `kl.backward()` FOLLOWED BY `nll.backward()` needs `retain_graph=True` on the first call (the other order,
    and the usual single `(nll + c * kl).backward()`, need nothing)."""
    try:
        return ctx.saved_tensors
    except RuntimeError as e:
        if getattr(ctx, "kl_only_ran", False):
            raise RuntimeError(
                "the layer's KL term was back-propagated on its own BEFORE the data term; the fused KL shares the layer's "
                "autograd node, so that pass freed the saved activations: call nll.backward() first, use one combined "
                "backward, or pass retain_graph=True to the KL backward") from e
        raise


# Data-parallel hooks (cplxmodule_amd.dp.BucketHook): the linear layers' backward writes the parameter gradients straight
# into their all-reduce bucket (`grad_buffer`) and announces them (`early_ready`) BEFORE its input-gradient GEMMs, so the
# RCCL all-reduce of a full bucket overlaps them.  Hooks are found PER PARAMETER (`register_dp_hook`: one entry per
# parameter of a DataParallel wrapper), so several wrapped models live in one process; `dp_hook` is the process-wide
# fallback consulted for parameters no wrapper registered (the first wrapper installs itself there; tests install fakes).
dp_hook = None
_dp_hooks = {}            # id(parameter) -> weak reference to its wrapper's hook (the hook keeps the parameter alive)


def register_dp_hook(hook, params):
    import weakref
    ref = weakref.ref(hook)
    for p in params:
        _dp_hooks[id(p)] = ref


def unregister_dp_hook(hook):
    for k in [k for k, r in _dp_hooks.items() if r() is hook or r() is None]:
        del _dp_hooks[k]


def hook_of(param):
    """The data-parallel hook responsible for `param` (None: no wrapper)."""
    if _dp_hooks and param is not None:
        r = _dp_hooks.get(id(param))
        if r is not None:
            h = r()
            if h is not None:
                return h
            del _dp_hooks[id(param)]
    return dp_hook


def grad_buffer(param):
    """float32 storage for the gradient of `param`: a fresh view of its data-parallel bucket slice when a
    DataParallel wrapper holds it (the all-reduce then needs no copy), a new tensor otherwise."""
    h = hook_of(param)
    if h is not None:
        v = h.view_for(param)
        if v is not None:
            return v
    return torch.empty(param.shape, dtype=torch.float32, device=param.device)


def _announce(*params):
    if dp_hook is None and not _dp_hooks:
        return
    by_hook = {}
    for p in params:
        h = hook_of(p) if p is not None else None
        if h is not None:
            by_hook.setdefault(id(h), (h, []))[1].append(p)
    for h, ps in by_hook.values():
        h.early_ready(*ps)


class CplxLinearFn(torch.autograd.Function):
    """cplx.linear (cplxmodule/cplx.py:634-648) + its backward.  `mask` (float32, weight
    shape, not differentiated): the masked layers' `weight * mask` (nn/masked/complex.py:33-82) folded into
    the operand preparation (one pass: multiply + conversion to the activation dtype) and, in backward,
    into the epilogue of the weight-gradient GEMM."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, algo=0, mask=None):
        require_device(xr, xi, wr, wi, br, bi, mask)
        I, O = wr.shape[1], wr.shape[0]
        x2r, x2i = xr.reshape(-1, I).contiguous(), xi.reshape(-1, I).contiguous()
        bias = None if br is None else (_f32(_c(br)), _f32(_c(bi)))
        # Gauss 3M exists for the bf16 MFMA path only; float32 always runs the exact 4M kernel
        ctx.algo = algo = algo if _is_bf16(x2r) else 0
        if mask is not None:
            mask = _f32(_c(mask.expand_as(wr)))
            wmr, wmi = mask_mul(wr, wi, mask, out_dtype=x2r.dtype)
        else:
            wmr, wmi = _c(wr), _c(wi)
        ctx.mode = x3.get_fp32_mode()        # (the backward runs on an autograd thread: it follows this pass's choice)
        xs = _Pieces(x2r, x2i)
        yr, yi, ctx.wc = _cplx_linear_fwd(x2r, x2i, wmr, wmi, bias, algo, mode=ctx.mode, xs=xs)
        ctx.kind = x3.take(O, I, x2r.shape[0], mode=ctx.mode) if xs.v is not None else None     # (the weight gradient's)
        keep = xs.saved() if (_X3_SAVE and ctx.kind and xs.v[0].kind == ctx.kind) else (None, None, None)
        # (xr, xi as given: the create_graph backward needs graph-connected tensors, and x2r / x2i are views or copies made
        #  here, without history; for a contiguous input they share its storage)
        ctx.save_for_backward(x2r, x2i, wr, wi, mask, *keep, xr, xi)
        ctx.has_bias = br is not None
        ctx.lead = xr.shape[:-1]
        return yr.view(*ctx.lead, O), yi.view(*ctx.lead, O)

    @staticmethod
    def backward(ctx, gr, gi):
        x2r, x2i, wr, wi, mask, xsr, xsi, xsc, xr0, xi0 = ctx.saved_tensors
        O, I = wr.shape
        need = ctx.needs_input_grad
        dxr = dxi = dwr = dwi = dbr = dbi = None
        if torch.is_grad_enabled():
            # create_graph=True (gradient penalties, Hessian-vector products): the reference's linear is a composition of
            # differentiable torch ops (cplx.py:634-648), so its backward is differentiable too.  Same products, spelled
            # through THIS Function on conjugated / transposed operands -- the same kernels compute every order:
            #   dX = G conj(W) = linear(G, conj(W)^T),  dW = G^T conj(X) = linear(G^T, conj(X)^T),  db = sum_b G
            lin = lambda ar, ai, br, bi: CplxLinearFn.apply(ar, ai, br, bi, None, None, 0, None)  # noqa: E731
            G = (gr.reshape(-1, O), gi.reshape(-1, O))
            wmr, wmi = (wr, wi) if mask is None else (wr * mask, wi * mask)
            if need[0] or need[1]:
                dxr, dxi = lin(G[0], G[1], wmr.t(), -wmi.t())
                dxr, dxi = dxr.reshape(*ctx.lead, I), dxi.reshape(*ctx.lead, I)
            if need[2] or need[3]:
                dwr, dwi = lin(G[0].t(), G[1].t(), xr0.reshape(-1, I).t(), -xi0.reshape(-1, I).t())
                if mask is not None:
                    dwr, dwi = dwr * mask, dwi * mask
                dwr, dwi = dwr.to(wr.dtype), dwi.to(wi.dtype)
            if ctx.has_bias and (need[4] or need[5]):
                dbr, dbi = G[0].float().sum(0), G[1].float().sum(0)
            return dxr, dxi, dwr, dwi, dbr, dbi, None, None
        g2r, g2i = gr.reshape(-1, O).contiguous(), gi.reshape(-1, O).contiguous()
        gs = _Pieces(g2r, g2i)               # (float32 split products: the pieces of G serve dW and dX)
        xs = _Pieces(x2r, x2i, made=_Pieces.restored(ctx.kind, (xsr, xsi), xsc))
        # parameter gradients first (into their data-parallel bucket, announced before the dX GEMM)
        if need[2] or need[3]:
            dwr, dwi = grad_buffer(wr), grad_buffer(wi)
            _cplx_linear_dw(g2r, g2i, x2r, x2i, out=(dwr, dwi), algo=ctx.algo, emul=mask, mode=ctx.mode, gs=gs, xs=xs)
            _announce(wr, wi)
        if ctx.has_bias and (need[4] or need[5]):
            dbr, dbi = colsum2(g2r, g2i)
        if need[0] or need[1]:
            dxr, dxi = _cplx_linear_dx(g2r, g2i, ctx.wc[0], ctx.wc[1], x2r.dtype, ctx.algo, mode=ctx.mode, gs=gs)
            dxr, dxi = dxr.view(*ctx.lead, I), dxi.view(*ctx.lead, I)
        return dxr, dxi, dwr, dwi, dbr, dbi, None, None


def _kl_recompute(ctx):
    """The (unscaled) KL gradients of a layer once more, for a backward pass that finds its forward-pass buffers consumed
    (they hold totals by then: a second pass through a retained graph, or the KL-only pass after the data pass under a
    data-parallel hook).  The recomputation reads the CURRENT parameters: refuse when they changed since the forward pass
    (optimizer.step() with a retained graph would otherwise give silently wrong numbers -- ADVICE r4)."""
    if tuple(t._version for t in ctx.kl_params) != ctx.kl_versions:
        raise RuntimeError("a parameter of this layer was modified in place between its forward pass and a "
                           "backward pass that has to recompute the KL gradients (e.g. optimizer.step() "
                           "with a retained graph): the recomputation would use the NEW values")
    if len(ctx.kl_params) == 3:
        wr, wi, ls2 = ctx.kl_params
        return kl_fwd_bwd(ctx.kl_kind, wr, wi, ls2)[1:]
    w, ls2 = ctx.kl_params
    r = kl_fwd_bwd(ctx.kl_kind, w, None, ls2)
    return (r[1], r[2])


class CplxLinearLRTFn(torch.autograd.Function):
    """CplxLinearGaussian.forward in training mode (nn/relevance/complex/base.py:43-56):
    mu GEMM + variance GEMM + noise injection, backward per SURVEY A.2.

    With `kl_kind` the KL term of the layer rides along (third output = sum(penalty), complex/vd.py:95-99
    / ard.py:39): for bf16 activations ONE kernel reads the float32 parameters and writes the bf16 GEMM
    operands, the KL total and the unscaled KL gradients; the backward then finishes
    `dW = dW_data + g_kl * dW_kl` inside the epilogues of the weight-gradient GEMMs (device scalar
    g_kl), so neither autograd's gradient accumulation passes nor separate cast / exp / KL passes run."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i, seed, offset, kl_kind=None):
        require_device(xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i)
        ctx.set_materialize_grads(False)
        O, I = wr.shape
        x2r, x2i = xr.reshape(-1, I).contiguous(), xi.reshape(-1, I).contiguous()
        B = x2r.shape[0]
        bias = None if br is None else (_f32(_c(br)), _f32(_c(bi)))
        wrc, wic, ls2c = _c(wr), _c(wi), _c(ls2)
        kl = ctx.klg = None
        if kl_kind is not None:
            # the (unscaled) KL gradients are written now -- into the parameters' data-parallel bucket
            # slices when there is one -- and the data gradients are added to them in backward
            ctx.klg = (grad_buffer(ls2), grad_buffer(wr), grad_buffer(wi))
            # under a data-parallel hook these are views of the parameters' bucket slices: whatever a backward pass
            # returns is copied into (or IS) that storage, so the pending KL gradients survive only the pass that
            # consumes them (nll.backward(); (c * kl).backward() returned the data gradient as the KL one)
            ctx.klg_shared = hook_of(ls2) is not None
        # float32 layers: the three products of the forward (and the five of the backward) on split bf16 operands where
        # x3.take says so -- float32-level results at the bf16 pipe's rate / 6 instead of the float32 MFMA's
        ctx.mode = mode = x3.get_fp32_mode()
        use3 = x3.take(B, O, I, x2r, x2i, wrc, wic, ls2c, mode=mode) if x2r.dtype == torch.float32 else None
        if _prep_ok(x2r, wrc, wic, ls2c):
            wcr, wci, S, kl, _ = prep_kl(kl_kind, wrc, wic, ls2c, kl_kind is not None, ctx.klg)
        else:
            wcr, wci = cast(wrc, x2r.dtype), cast(wic, x2r.dtype)
            S = None if use3 else exp(ls2c, out_dtype=x2r.dtype)                # [O,I]
            if kl_kind is not None:
                kl = torch.empty((), dtype=torch.float32, device=x2r.device)
                g = ctx.klg
                call("cplxamd_vd_kl_fwd_bwd", ptr(_f32(wrc)), ptr(_f32(wic)), ptr(_f32(ls2c)), _lib.KL_KINDS[kl_kind],
                     1.0, ptr(kl), ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(_ws(x2r.device)), O * I, stream_ptr())
        ctx.wc, ctx.S = (wcr, wci), S
        keep = (None,) * 5
        ctx.kind = None
        if use3:
            xs, xa = _Pieces(x2r, x2i), _Pieces(x2r, x2i, abs2=True)
            mur, mui, _ = _cplx_linear_fwd(x2r, x2i, wcr, wci, bias, mode=mode, xs=xs)
            a = None                                         # |x|^2 exists as 16-bit pieces only
            s2 = x3.gemm_nn(xa.get(use3), (x3.split(ls2c, x3.SPLIT_B, op=x3.OP_EXP, kind=use3),), B, O, I)
            ctx.kind = x3.take(O, I, B, mode=mode)           # the weight gradients' arithmetic (K = batch)
            if _X3_SAVE and ctx.kind == use3:                # they will read the same pieces
                keep = (*xs.saved(), *xa.saved())
        else:
            mur, mui = cgemm(x2r, x2i, (I, 1), wcr, wci, (I, 1), B, O, I, bias=bias, out_dtype=x2r.dtype)
            a = abs2(x2r, x2i)                                   # [B,I], activation dtype
            # [B,O]; bf16 layers keep the variance in bf16 (2 bytes per output less in the GEMM epilogue and in both noise
            # passes; sigma enters y = mu + eps sigma, itself rounded to bf16, with a relative error of 2^-10)
            s2 = rgemm(a, (I, 1), S, (I, 1), B, O, I, out_dtype=_s2_dtype(x2r))
        eps = None
        if eps_r is not None:
            eps = (eps_r.reshape(B, O), eps_i.reshape(B, O))
        yr, yi = reparam_fwd(mur, mui, s2, eps, seed, offset, inplace=True)
        ctx.save_for_backward(x2r, x2i, wr, wi, ls2, s2, a, eps_r, eps_i, br, bi, *keep)
        ctx.has_bias = br is not None
        ctx.lead, ctx.seed, ctx.offset = xr.shape[:-1], seed, offset
        ctx.kl_kind = kl_kind
        ctx.kl_params = (wr, wi, ls2) if kl_kind is not None else None    # (leaves: no cycle through ctx)
        ctx.kl_versions = tuple(t._version for t in ctx.kl_params) if kl_kind is not None else None
        return yr.view(*ctx.lead, O), yi.view(*ctx.lead, O), kl

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi, gkl=None):
        need = ctx.needs_input_grad
        dxr = dxi = dwr = dwi = dbr = dbi = dls2 = None
        if gr is None and gi is None:
            # only the KL term reached the loss (e.g. `nll.backward(); (c * sum(penalties(model))).backward()`, the
            # reference's two-call pattern): needs no saved tensor, so it also works after the data backward freed them
            if gkl is not None and ctx.kl_kind is not None:
                if ctx.klg is None:          # the buffers hold totals by now: redo the KL part
                    ctx.klg, ctx.klg_shared = _kl_recompute(ctx), False
                dls2, dwr, dwi = (_scaled(gkl, t) for t in ctx.klg)
                if ctx.klg_shared:
                    ctx.klg = None           # the hook overwrites the bucket slices with what this pass returns
            ctx.kl_only_ran = True
            return dxr, dxi, dwr, dwi, dbr, dbi, dls2, None, None, None, None, None
        x2r, x2i, wr, wi, ls2, s2, a, eps_r, eps_i, br, bi, xsr, xsi, xsc, xsa, xac = _saved(ctx)
        O, I = wr.shape
        B = x2r.shape[0]
        klg = None
        if gkl is not None and ctx.kl_kind is not None:
            # (a second backward through a retained graph: the buffers hold totals by then, redo the KL part)
            klg = _kl_recompute(ctx) if ctx.klg is None else ctx.klg
        g2r = torch.zeros(B, O, dtype=x2r.dtype, device=x2r.device) if gr is None else gr.reshape(B, O).contiguous()
        g2i = torch.zeros(B, O, dtype=x2r.dtype, device=x2r.device) if gi is None else gi.reshape(B, O).contiguous()
        eps = None if eps_r is None else (eps_r.reshape(B, O), eps_i.reshape(B, O))
        dt = x2r.dtype
        # parameter gradients first: under data parallelism their bucket's all-reduce overlaps dX
        want_w, want_b = need[2] or need[3], ctx.has_bias and (need[4] or need[5])
        if want_b:          # the bias gradient (column sums of G) rides in the pass that reads G for d s2
            gs2, dbr, dbi = reparam_bwd(g2r, g2i, s2, eps, ctx.seed, ctx.offset, out_dtype=dt,
                                        bias_sums=(B, O, (grad_buffer(br), grad_buffer(bi))))
        else:
            gs2 = reparam_bwd(g2r, g2i, s2, eps, ctx.seed, ctx.offset, out_dtype=dt)
        ls2c = _c(ls2)
        # dW = G^T conj(X) + g_kl * dW_kl and dls2 = (gs2^T |x|^2) exp(ls2) + g_kl * dls2_kl.  While ctx.klg holds the
        # UNSCALED KL gradients of the forward pass (under data parallelism: in the parameters' bucket slices, the very
        # storage grad_buffer() hands out), the GEMM epilogues finish the sum in place, per tensor; a tensor whose
        # gradient is not wanted is left alone.  (ctx.klg are views nobody else holds once ctx dies, so autograd
        # adopts them without a copy.)
        fused = klg is not None and klg is ctx.klg
        # a data-only backward with the KL buffers still pending (the two-call pattern) must not write into them
        own = (lambda p: torch.empty(p.shape, dtype=torch.float32, device=p.device)) if (ctx.klg is not None and not fused) \
            else grad_buffer
        # float32 split products: pieces made once per pass, shared by the products that read them
        m3 = ctx.mode
        gs, g2s, xa = _Pieces(g2r, g2i), _Pieces(gs2), None
        xs = _Pieces(x2r, x2i, made=_Pieces.restored(ctx.kind, (xsr, xsi), xsc))
        if a is None:                                        # the forward ran on pieces of |x|^2
            if x3.take(O, I, B, gs2, mode=m3):
                xa = _Pieces(x2r, x2i, abs2=True, made=_Pieces.restored(ctx.kind, (xsa,), xac))
            else:
                a = abs2(x2r, x2i)
        if want_w:
            if fused:
                dwr, dwi = klg[1], klg[2]
                _cplx_linear_dw(g2r, g2i, x2r, x2i, out=(dwr, dwi), accumulate=True, beta=gkl, mode=m3, gs=gs, xs=xs)
            else:
                dwr, dwi = own(wr), own(wi)
                _cplx_linear_dw(g2r, g2i, x2r, x2i, out=(dwr, dwi), mode=m3, gs=gs, xs=xs)
                if klg is not None:
                    dwr.add_(klg[1] * gkl)
                    dwi.add_(klg[2] * gkl)
            # two thirds of the layer's gradient bytes are final here: their buckets' all-reduces may start under the
            # variance weight gradient already (0.25 ms earlier than behind the announcement below)
            if _EARLY_W:
                _announce(wr, wi)
        if need[6]:
            if fused:
                dls2 = klg[0]
                _real_linear_dw(gs2, a, emul=ls2c, emul_exp=True, out=dls2, accumulate=True, beta=gkl, mode=m3, gs=g2s, xs=xa)
            else:
                dls2 = own(ls2)
                _real_linear_dw(gs2, a, emul=ls2c, emul_exp=True, out=dls2, mode=m3, gs=g2s, xs=xa)  # (gs2^T a) * exp(ls2)
                if klg is not None:
                    dls2.add_(klg[0] * gkl)
        if fused or getattr(ctx, "klg_shared", False):
            ctx.klg = None                                   # consumed (or about to be overwritten by the hook's copy): the buffers hold totals now
        if not _EARLY_W:
            _announce(wr if dwr is not None else None, wi if dwi is not None else None)
        _announce(ls2 if dls2 is not None else None, br if dbr is not None else None, bi if dbi is not None else None)
        if need[0] or need[1]:
            if ctx.S is None:                                # gs2 . exp(ls2) -> [B,I], sigma^2 formed in the split pass
                ga = _real_linear_dx(gs2, ls2c, dt, mode=m3, gs=g2s, w_exp=True)
            else:
                ga = _real_linear_dx(gs2, ctx.S, dt)
            dxr, dxi = _cplx_lrt_dx(g2r, g2i, ctx.wc[0], ctx.wc[1], x2r, x2i, ga, mode=m3, gs=gs)   # G conj(W) + 2 x ga
            dxr, dxi = dxr.view(*ctx.lead, I), dxi.view(*ctx.lead, I)
        return dxr, dxi, dwr, dwi, dbr, dbi, dls2, None, None, None, None, None


class RealLinearFn(torch.autograd.Function):
    """F.linear on the real GEMM kernel (the mean of LinearGaussian, real/base.py:44); `mask` as in
    CplxLinearFn (LinearMasked, nn/masked/real.py:25-71)."""

    @staticmethod
    def forward(ctx, x, w, b, mask=None):
        require_device(x, w, b, mask)
        O, I = w.shape
        x2 = x.reshape(-1, I).contiguous()
        if mask is not None:
            mask = _f32(_c(mask.expand_as(w)))
            wm, _ = mask_mul(w, None, mask, out_dtype=x2.dtype)
        else:
            wm = _c(w)
        ctx.mode = x3.get_fp32_mode()
        xs = _Pieces(x2)
        y, wm = _real_linear_fwd(x2, wm, _c(b), mode=ctx.mode, xs=xs)
        ctx.wm = wm
        ctx.kind = x3.take(O, I, x2.shape[0], mode=ctx.mode) if xs.v is not None else None
        keep = xs.saved() if (_X3_SAVE and ctx.kind and xs.v[0].kind == ctx.kind) else (None, None)
        ctx.save_for_backward(x2, w, mask, *keep, x)         # (x as given: see CplxLinearFn.forward)
        ctx.has_bias, ctx.lead = b is not None, x.shape[:-1]
        return y.view(*ctx.lead, O)

    @staticmethod
    def backward(ctx, g):
        x2, w, mask, xsv, xsc, x0 = ctx.saved_tensors
        O, I = w.shape
        need = ctx.needs_input_grad
        dx = dw = db = None
        if torch.is_grad_enabled():          # create_graph=True: through this Function itself (see CplxLinearFn.backward)
            lin = lambda a, b: RealLinearFn.apply(a, b, None, None)  # noqa: E731
            G = g.reshape(-1, O)
            wm = w if mask is None else w * mask
            if need[0]:
                dx = lin(G, wm.t()).reshape(*ctx.lead, I)
            if need[1]:
                dw = lin(G.t(), x0.reshape(-1, I).t())
                dw = (dw if mask is None else dw * mask).to(w.dtype)
            if ctx.has_bias and need[2]:
                db = G.float().sum(0)
            return dx, dw, db, None
        g2 = g.reshape(-1, O).contiguous()
        gs = _Pieces(g2)
        if need[1]:
            dw = grad_buffer(w)
            _real_linear_dw(g2, x2, emul=mask, out=dw, mode=ctx.mode, gs=gs,
                            xs=None if xsv is None else _Pieces(x2, made=_Pieces.restored(ctx.kind, (xsv,), xsc)))
            _announce(w)
        if ctx.has_bias and need[2]:
            db = colsum(g2)
        if need[0]:
            dx = _real_linear_dx(g2, ctx.wm, x2.dtype, mode=ctx.mode, gs=gs).view(*ctx.lead, I)
        return dx, dw, db, None


class MaskMulFn(torch.autograd.Function):
    """(wr * mask, wi * mask) as one kernel (wi None: real weight); the conv / bilinear masked layers."""

    @staticmethod
    def forward(ctx, wr, wi, mask):
        mask = _f32(_c(mask.expand_as(wr)))
        ctx.save_for_backward(mask)
        ctx.cplx = wi is not None
        our, oui = mask_mul(wr, wi, mask)
        return (our, oui) if ctx.cplx else our

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi=None):
        (mask,) = ctx.saved_tensors
        dr, di = mask_mul(gr, gi if ctx.cplx else None, mask)
        return dr, di, None


class RealLinearLRTFn(torch.autograd.Function):
    """LinearGaussian.forward in training mode (nn/relevance/real/base.py:43-49); `kl_kind` as in
    CplxLinearLRTFn (second output = sum(penalty), real/vd.py:54-76 / ard.py:39)."""

    @staticmethod
    def forward(ctx, x, w, b, ls2, eps, seed, offset, kl_kind=None):
        require_device(x, w, b, ls2, eps)
        ctx.set_materialize_grads(False)
        O, I = w.shape
        x2 = x.reshape(-1, I).contiguous()
        B = x2.shape[0]
        wc_, ls2c = _c(w), _c(ls2)
        kl = ctx.klg = None
        if kl_kind is not None:
            ctx.klg = (grad_buffer(ls2), grad_buffer(w))
            ctx.klg_shared = hook_of(ls2) is not None        # (see CplxLinearLRTFn.forward)
        ctx.mode = mode = x3.get_fp32_mode()                 # (see CplxLinearLRTFn.forward)
        use3 = x3.take(B, O, I, x2, wc_, ls2c, mode=mode) if x2.dtype == torch.float32 else None
        if _prep_ok(x2, wc_, ls2c):
            wb, _, S, kl, _ = prep_kl(kl_kind, wc_, None, ls2c, kl_kind is not None,
                                      None if ctx.klg is None else (*ctx.klg, None))
        else:
            wb, S = cast(wc_, x2.dtype), (None if use3 else exp(ls2c, out_dtype=x2.dtype))
            if kl_kind is not None:
                kl = torch.empty((), dtype=torch.float32, device=x2.device)
                call("cplxamd_vd_kl_fwd_bwd", ptr(_f32(wc_)), None, ptr(_f32(ls2c)), _lib.KL_KINDS[kl_kind], 1.0,
                     ptr(kl), ptr(ctx.klg[0]), ptr(ctx.klg[1]), None, ptr(_ws(x2.device)), O * I, stream_ptr())
        ctx.wb, ctx.S = wb, S
        keep = (None,) * 4
        ctx.kind = None
        if use3:
            xs, xa = _Pieces(x2), _Pieces(x2, abs2=True)
            mu, _ = _real_linear_fwd(x2, wb, _c(b), mode=mode, xs=xs)
            a = None
            s2 = x3.gemm_nn(xa.get(use3), (x3.split(ls2c, x3.SPLIT_B, op=x3.OP_EXP, kind=use3),), B, O, I)
            ctx.kind = x3.take(O, I, B, mode=mode)
            if _X3_SAVE and ctx.kind == use3:
                keep = (*xs.saved(), *xa.saved())
        else:
            mu = rgemm(x2, (I, 1), wb, (I, 1), B, O, I, bias=_c(b), out_dtype=x2.dtype)
            a = abs2(x2)
            s2 = rgemm(a, (I, 1), S, (I, 1), B, O, I, out_dtype=_s2_dtype(x2))
        e = None if eps is None else eps.reshape(B, O)
        y, _ = reparam_fwd(mu, None, s2, e, seed, offset, inplace=True)
        ctx.save_for_backward(x2, w, ls2, s2, a, eps, b, *keep)
        ctx.has_bias, ctx.lead, ctx.seed, ctx.offset = b is not None, x.shape[:-1], seed, offset
        ctx.kl_kind = kl_kind
        ctx.kl_params = (w, ls2) if kl_kind is not None else None
        ctx.kl_versions = tuple(t._version for t in ctx.kl_params) if kl_kind is not None else None
        return y.view(*ctx.lead, O), kl

    @staticmethod
    @once_differentiable
    def backward(ctx, g, gkl=None):
        need = ctx.needs_input_grad
        dx = dw = db = dls2 = None
        if g is None:                                        # only the KL term reached the loss (see CplxLinearLRTFn)
            if gkl is not None and ctx.kl_kind is not None:
                if ctx.klg is None:
                    ctx.klg, ctx.klg_shared = _kl_recompute(ctx), False
                dls2, dw = _scaled(gkl, ctx.klg[0]), _scaled(gkl, ctx.klg[1])
                if ctx.klg_shared:
                    ctx.klg = None
            ctx.kl_only_ran = True
            return dx, dw, db, dls2, None, None, None, None
        x2, w, ls2, s2, a, eps, b, xsv, xsc, xsa, xac = _saved(ctx)
        O, I = w.shape
        B = x2.shape[0]
        klg = None
        if gkl is not None and ctx.kl_kind is not None:
            klg = _kl_recompute(ctx) if ctx.klg is None else ctx.klg   # (second backward through a retained graph)
        g2 = g.reshape(B, O).contiguous()
        dt = x2.dtype
        e = None if eps is None else eps.reshape(B, O)
        if ctx.has_bias and need[2]:
            gs2, db, _ = reparam_bwd(g2, None, s2, e, ctx.seed, ctx.offset, out_dtype=dt,
                                     bias_sums=(B, O, (grad_buffer(b), None)))
        else:
            gs2 = reparam_bwd(g2, None, s2, e, ctx.seed, ctx.offset, out_dtype=dt)
        ls2c = _c(ls2)
        fused = klg is not None and klg is ctx.klg           # per tensor, in place (see CplxLinearLRTFn.backward)
        own = (lambda p: torch.empty(p.shape, dtype=torch.float32, device=p.device)) if (ctx.klg is not None and not fused) \
            else grad_buffer
        m3 = ctx.mode
        gs, g2s, xa = _Pieces(g2), _Pieces(gs2), None        # (see CplxLinearLRTFn.backward)
        xs = None if xsv is None else _Pieces(x2, made=_Pieces.restored(ctx.kind, (xsv,), xsc))
        if a is None:
            if x3.take(O, I, B, gs2, mode=m3):
                xa = _Pieces(x2, abs2=True, made=_Pieces.restored(ctx.kind, (xsa,), xac))
            else:
                a = abs2(x2)
        if need[1]:
            if fused:
                dw = klg[1]
                _real_linear_dw(g2, x2, out=dw, accumulate=True, beta=gkl, mode=m3, gs=gs, xs=xs)
            else:
                dw = own(w)
                _real_linear_dw(g2, x2, out=dw, mode=m3, gs=gs, xs=xs)
                if klg is not None:
                    dw.add_(klg[1] * gkl)
        if need[3]:
            if fused:
                dls2 = klg[0]
                _real_linear_dw(gs2, a, emul=ls2c, emul_exp=True, out=dls2, accumulate=True, beta=gkl, mode=m3, gs=g2s, xs=xa)
            else:
                dls2 = own(ls2)
                _real_linear_dw(gs2, a, emul=ls2c, emul_exp=True, out=dls2, mode=m3, gs=g2s, xs=xa)
                if klg is not None:
                    dls2.add_(klg[0] * gkl)
        if fused or getattr(ctx, "klg_shared", False):
            ctx.klg = None
        _announce(ls2 if dls2 is not None else None, w if dw is not None else None, b if db is not None else None)
        if need[0]:
            if ctx.S is None:
                ga = _real_linear_dx(gs2, ls2c, dt, mode=m3, gs=g2s, w_exp=True)
            else:
                ga = _real_linear_dx(gs2, ctx.S, dt)
            dx = _real_lrt_dx(g2, ctx.wb if _is_bf16(g2) else _c(w), x2, ga, mode=m3, gs=gs)
            dx = dx.view(*ctx.lead, I)
        return dx, dw, db, dls2, None, None, None, None


# ------------------------------------------------------------------------------------------ #
#  bilinear layers: GEMM over the second input, then the reduction kernels of bilinear.hip     #
# ------------------------------------------------------------------------------------------ #
def bilinear_reduce_fwd(u, t, bias, B, O, I1, conj):
    """y[b,o] = sum_i conj?(u[b,i]) t[b,o,i] + bias[o]; u, t, bias: (re, im) pairs, im None = real."""
    (ur, ui), (tr, ti) = u, t
    require_device(ur, ui, tr, ti)
    yr = torch.empty(B, O, dtype=tr.dtype, device=tr.device)
    yi = None if ti is None else torch.empty_like(yr)
    br, bi = (None, None) if bias is None else bias
    call("cplxamd_bilinear_reduce_fwd", ptr(ur), ptr(ui), ptr(tr), ptr(ti), ptr(br), ptr(bi), ptr(yr),
         ptr(yi), B, O, I1, int(conj), dtype_code(tr), stream_ptr())
    return yr, yi


def bilinear_reduce_bwd(u, t, g, B, O, I1, conj, need_u=True, need_t=True):
    """-> (dx1_r, dx1_i), (dt_r, dt_i) of bilinear_reduce_fwd (pairs of None where not needed)."""
    (ur, ui), (tr, ti), (gr, gi) = u, t, g
    require_device(ur, ui, tr, ti, gr, gi)
    cplx_ = ui is not None
    dur = torch.empty_like(ur) if need_u else None
    dui = torch.empty_like(ur) if need_u and cplx_ else None
    dtr = torch.empty(B, O * I1, dtype=gr.dtype, device=gr.device) if need_t else None
    dti = torch.empty_like(dtr) if need_t and cplx_ else None
    call("cplxamd_bilinear_reduce_bwd", ptr(ur), ptr(ui), ptr(tr), ptr(ti), ptr(gr), ptr(gi), ptr(dur),
         ptr(dui), ptr(dtr), ptr(dti), B, O, I1, int(conj), dtype_code(gr), stream_ptr())
    return (dur, dui), (dtr, dti)


def _bil_forward(ctx, x1, x2, w, bias, conj, ls2, eps, seed, offset):
    """Shared forward of the four bilinear Functions.  x1, x2, w, bias, eps: (re, im) pairs with
    im None for real layers; ls2 None = no noise (plain layer / eval mode)."""
    cplx_ = x1[1] is not None
    O, I1, I2 = w[0].shape
    lead = x1[0].shape[:-1]
    f1 = tuple(None if p is None else p.reshape(-1, I1).contiguous() for p in x1)
    f2 = tuple(None if p is None else p.reshape(-1, I2).contiguous() for p in x2)
    B, dt = f1[0].shape[0], f1[0].dtype
    wv = tuple(None if p is None else _c(p).view(O * I1, I2) for p in w)
    fb = None if bias is None else tuple(None if p is None else _f32(_c(p)) for p in bias)
    if cplx_:
        tr, ti, wc = _cplx_linear_fwd(f2[0], f2[1], wv[0], wv[1], None)
    else:
        wc = (cast(wv[0], dt), None)
        tr, ti = rgemm(f2[0], (I2, 1), wc[0], (I2, 1), B, O * I1, I2, out_dtype=dt), None
    yr, yi = bilinear_reduce_fwd(f1, (tr, ti), fb, B, O, I1, conj)
    s2 = a1 = a2 = tv = None
    if ls2 is not None:
        a1, a2 = abs2(f1[0], f1[1], out_dtype=torch.float32), abs2(f2[0], f2[1])
        S = exp(_c(ls2), out_dtype=dt).view(O * I1, I2)
        tv = rgemm(a2, (I2, 1), S, (I2, 1), B, O * I1, I2)              # float32 [B, O*I1]
        s2, _ = bilinear_reduce_fwd((a1, None), (tv, None), None, B, O, I1, False)
        e = None
        if eps is not None and eps[0] is not None:
            e = tuple(None if p is None else p.reshape(B, O) for p in eps)
            e = e if cplx_ else e[0]
        yr, yi = reparam_fwd(yr, yi, s2, e, seed, offset, inplace=True)
    ctx.bil = dict(f1=f1, f2=f2, t=(tr, ti), wc=wc, s2=s2, a1=a1, a2=a2, tv=tv, ls2=ls2, eps=eps,
                   seed=seed, offset=offset, conj=conj, lead=lead, dims=(B, O, I1, I2),
                   has_bias=bias is not None)
    return (yr.view(*lead, O), None if yi is None else yi.view(*lead, O))


def _bil_backward(ctx, g, need_x1, need_x2, need_w, need_b, need_ls2):
    """-> dx1, dx2, dw, db (pairs), dls2.  SURVEY A.1 / A.2 with the weight seen as [(o,i), j]."""
    st = ctx.bil
    B, O, I1, I2 = st["dims"]
    f1, f2, t, wc = st["f1"], st["f2"], st["t"], st["wc"]
    cplx_ = f1[1] is not None
    dt = f1[0].dtype
    g = tuple(None if p is None else p.reshape(B, O).contiguous() for p in g)
    dx1, dT = bilinear_reduce_bwd(f1, t, g, B, O, I1, st["conj"], need_u=need_x1,
                                  need_t=need_x2 or need_w)
    dx2 = dw = (None, None)
    db = (None, None)
    dls2 = None
    if need_x2:
        if cplx_:
            dx2 = _cplx_linear_dx(dT[0], dT[1], wc[0], wc[1], dt)
        else:
            dx2 = (_real_linear_dx(dT[0], wc[0], dt), None)
    if need_w:
        if cplx_:
            dw = _cplx_linear_dw(dT[0], dT[1], f2[0], f2[1])
        else:
            dw = (_real_linear_dw(dT[0], f2[0]), None)
        dw = tuple(None if p is None else p.view(O, I1, I2) for p in dw)
    if need_b and st["has_bias"]:
        db = tuple(None if p is None else colsum(p) for p in g)
    if st["ls2"] is not None:
        e = st["eps"]
        if e is not None and e[0] is not None:
            e = tuple(None if p is None else p.reshape(B, O) for p in e)
            e = e if cplx_ else e[0]
        else:
            e = None
        gs2 = reparam_bwd(g[0], g[1], st["s2"], e, st["seed"], st["offset"])       # float32 [B,O]
        (da1, _), (dtv, _) = bilinear_reduce_bwd((st["a1"], None), (st["tv"], None), (gs2, None), B, O,
                                                 I1, False, need_u=need_x1, need_t=need_x2 or need_ls2)
        if need_x1:
            lrt_dx_accum(dx1[0], dx1[1], f1[0], f1[1], da1)
        ls2c = _c(st["ls2"])
        if need_x2:
            if _is_bf16(f2[0]):
                da2 = _real_linear_dx(cast(dtv, dt), exp(ls2c, out_dtype=dt).view(O * I1, I2), dt)
            else:
                da2 = _real_linear_dx(dtv, exp(ls2c).view(O * I1, I2), dt)
            lrt_dx_accum(dx2[0], dx2[1], f2[0], f2[1], da2)
        if need_ls2:
            dls2 = _real_linear_dw(dtv, st["a2"], emul=exp(ls2c).view(O * I1, I2)).view(O, I1, I2)
    lead = st["lead"]
    dx1 = tuple(None if p is None else p.view(*lead, I1) for p in dx1)
    dx2 = tuple(None if p is None else p.view(*lead, I2) for p in dx2)
    return dx1, dx2, dw, db, dls2


class CplxBilinearFn(torch.autograd.Function):
    """cplx.bilinear (cplxmodule/cplx.py:1062-1087); with `ls2` also the training-mode forward of
    CplxBilinearGaussian (nn/relevance/complex/base.py:73-84)."""

    @staticmethod
    def forward(ctx, x1r, x1i, x2r, x2i, wr, wi, br, bi, conj, ls2, eps_r, eps_i, seed, offset):
        require_device(x1r, x1i, x2r, x2i, wr, wi, br, bi, ls2, eps_r, eps_i)
        return _bil_forward(ctx, (x1r, x1i), (x2r, x2i), (wr, wi), None if br is None else (br, bi),
                            conj, ls2, (eps_r, eps_i), seed, offset)

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        n = ctx.needs_input_grad
        dx1, dx2, dw, db, dls2 = _bil_backward(ctx, (gr, gi), n[0] or n[1], n[2] or n[3], n[4] or n[5],
                                               n[6] or n[7], n[9])
        return dx1[0], dx1[1], dx2[0], dx2[1], dw[0], dw[1], db[0], db[1], None, dls2, None, None, None, None


class RealBilinearFn(torch.autograd.Function):
    """F.bilinear on the GEMM + reduction kernels; with `ls2` the training-mode forward of
    BilinearGaussian (nn/relevance/real/base.py:66-77)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b, ls2, eps, seed, offset):
        require_device(x1, x2, w, b, ls2, eps)
        y, _ = _bil_forward(ctx, (x1, None), (x2, None), (w, None), None if b is None else (b, None),
                            False, ls2, (eps, None), seed, offset)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n = ctx.needs_input_grad
        dx1, dx2, dw, db, dls2 = _bil_backward(ctx, (g, None), n[0], n[1], n[2], n[3], n[4])
        return dx1[0], dx2[0], dw[0], db[0], dls2, None, None, None


class Abs2Fn(torch.autograd.Function):
    """|x|^2 (xi None: x^2) as a differentiable op, for variance paths composed from several
    kernels (conv3d.py)."""

    @staticmethod
    def forward(ctx, xr, xi):
        xr, xi = _c(xr), _c(xi)
        ctx.save_for_backward(xr, xi)
        return abs2(xr, xi)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xr, xi = ctx.saved_tensors
        dxr = torch.zeros_like(xr)
        dxi = None if xi is None else torch.zeros_like(xi)
        lrt_dx_accum(dxr, dxi, xr, xi, _c(g))              # dx = 2 x g
        return dxr, dxi


class AbsFn(torch.autograd.Function):
    """abs(Cplx) (cplxmodule/cplx.py:183-192) with the subgradient 0 at z == 0 that the reference's
    stack + norm has (autograd through sqrt(re^2 + im^2) gives NaN there)."""

    @staticmethod
    def forward(ctx, zr, zi):
        require_device(zr, zi)
        ctx.fmt = fmt = _layout_of(zr)
        zr, zi = _cf(zr, fmt), _cf(zi, fmt)
        ctx.save_for_backward(zr, zi)
        return modulus(zr, zi)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        zr, zi = ctx.saved_tensors
        g = _cf(g, ctx.fmt)
        if g.dtype != zr.dtype:
            g = cast(g, zr.dtype)
        dzr, dzi = torch.empty_like(zr), torch.empty_like(zi)
        call("cplxamd_cplx_abs_bwd", ptr(g), ptr(zr), ptr(zi), ptr(dzr), ptr(dzi), zr.numel(),
             dtype_code(zr), stream_ptr())
        return dzr, dzi


class LogAlphaFn(torch.autograd.Function):
    """log_alpha = log_sigma2 - 2 log(abs(w) + 1e-12) (complex/base.py:27-31, real/base.py:23-26) on the
    exact-log kernel; backward: d log_sigma2 = g, d w = cplxamd_vd_log_alpha_bwd."""

    @staticmethod
    def forward(ctx, ls2, wr, wi):
        ctx.save_for_backward(wr, wi)
        return log_alpha(wr, wi, ls2).view_as(ls2)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        wr, wi = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_wr = g_wi = None
        if need[1] or (wi is not None and need[2]):
            g_wr, g_wi = log_alpha_bwd(g, wr, wi)
            g_wr = g_wr.view_as(wr)
            g_wi = None if g_wi is None else g_wi.view_as(wi)
        return (g if need[0] else None), g_wr, g_wi


class ExpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = exp(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g * ctx.saved_tensors[0]


class ReparamFn(torch.autograd.Function):
    """y = mu + eps sqrt(max(s2, 1e-8)) as a stand-alone differentiable op (the fused layers use
    the kernels directly); d mu = g, d s2 = reparam_bwd."""

    @staticmethod
    def forward(ctx, mu_r, mu_i, s2, eps_r, eps_i, seed, offset):
        require_device(mu_r, mu_i, s2, eps_r, eps_i)
        eps = None if eps_r is None else ((eps_r, eps_i) if mu_i is not None else eps_r)
        s2 = _c(s2)
        ctx.save_for_backward(s2, eps_r, eps_i)
        ctx.cplx, ctx.seed, ctx.offset = mu_i is not None, seed, offset
        yr, yi = reparam_fwd(mu_r, mu_i, s2, eps, seed, offset)
        return (yr.view_as(mu_r), None if yi is None else yi.view_as(mu_r))

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        s2, eps_r, eps_i = ctx.saved_tensors
        eps = None if eps_r is None else ((eps_r, eps_i) if ctx.cplx else eps_r)
        gs2 = reparam_bwd(gr, gi if ctx.cplx else None, s2, eps, ctx.seed, ctx.offset)
        return gr, (gi if ctx.cplx else None), gs2.view_as(s2), None, None, None, None


class PenaltyFn(torch.autograd.Function):
    """Elementwise KL penalty tensor (the `.penalty` property of the VD / ARD layers)."""

    @staticmethod
    def forward(ctx, kind, ls2, wr, wi):
        elem, _ = kl_fwd(kind, wr, wi, ls2, elementwise=True, total=False)
        ctx.kind = kind
        ctx.save_for_backward(ls2, wr, wi)
        return elem.view_as(ls2)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ls2, wr, wi = ctx.saved_tensors
        need = ctx.needs_input_grad[1:]
        g_ls2, g_wr, g_wi = kl_bwd(ctx.kind, wr, wi, ls2, g_elem=g.contiguous(), need=need)
        v = lambda t: None if t is None else t.view_as(ls2)  # noqa: E731
        return None, v(g_ls2), v(g_wr), v(g_wi)


class PenaltySumFn(torch.autograd.Function):
    """sum(penalty) as ONE fused elementwise + wavefront-shuffle reduction kernel; the backward
    re-reads the parameters and scales by the upstream scalar read on-device (no host sync)."""

    @staticmethod
    def forward(ctx, kind, ls2, wr, wi):
        _, tot = kl_fwd(kind, wr, wi, ls2, elementwise=False, total=True)
        ctx.kind = kind
        ctx.save_for_backward(ls2, wr, wi)
        return tot

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ls2, wr, wi = ctx.saved_tensors
        need = ctx.needs_input_grad[1:]
        g_ls2, g_wr, g_wi = kl_bwd(ctx.kind, wr, wi, ls2, g_scalar=g, need=need)
        v = lambda t: None if t is None else t.view_as(ls2)  # noqa: E731
        return None, v(g_ls2), v(g_wr), v(g_wi)


class ExpiFn(torch.autograd.Function):
    """torch_expi (cplxmodule/nn/relevance/complex/vd.py:15-44) without the host round trip."""

    @staticmethod
    def forward(ctx, x):
        require_device(x)
        xc = _f32(x.contiguous())
        y = torch.empty_like(xc)
        call("cplxamd_expi_fwd", ptr(xc), ptr(y), xc.numel(), stream_ptr())
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _f32(g.contiguous())
        gx = torch.empty_like(x)
        call("cplxamd_expi_bwd", ptr(g), ptr(x), ptr(gx), x.numel(), stream_ptr())
        return gx


# ------------------------------------------------------------------------------------------ #
#  layout converters, modReLU, complex dropout (csrc/layout.hip)        #
# ------------------------------------------------------------------------------------------ #
class DeinterleaveFn(torch.autograd.Function):
    """x[..., 2D] -> (re[..., D], im[..., D]) in one pass; backward = interleave of the gradients."""

    @staticmethod
    def forward(ctx, x):
        require_device(x)
        x = _al16(_c(x))
        shape = (*x.shape[:-1], x.shape[-1] // 2)
        re, im = torch.empty(shape, dtype=x.dtype, device=x.device), torch.empty(shape, dtype=x.dtype, device=x.device)
        call("cplxamd_deinterleave", ptr(x), ptr(re), ptr(im), re.numel(), dtype_code(x), stream_ptr())
        return re, im

    @staticmethod
    def backward(ctx, gr, gi):
        return InterleaveFn.apply(gr, gi)


class InterleaveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, re, im):
        require_device(re, im)
        re, im = _al16(_c(re)), _al16(_c(im))
        out = torch.empty((*re.shape[:-1], 2 * re.shape[-1]), dtype=re.dtype, device=re.device)
        call("cplxamd_interleave", ptr(re), ptr(im), ptr(out), re.numel(), dtype_code(re), stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return DeinterleaveFn.apply(g)


def _tau_args(tau, like, fmt=torch.contiguous_format):
    """(device pointer, value, numel) of a modReLU threshold: python float -> by value; 1-element
    tensor -> read on the device; anything else -> broadcast to the activation's shape (and storage layout)."""
    if not isinstance(tau, torch.Tensor):
        return None, float(tau), 0, None
    t = _f32(tau)
    if t.numel() == 1:
        return t, 0.0, 1, t
    t = _al16(t.expand(like.shape).contiguous(memory_format=fmt))
    return t, 0.0, t.numel(), t


class ModReluFn(torch.autograd.Function):
    """cplx.modrelu (cplxmodule/cplx.py:565-616) fused with its backward."""

    @staticmethod
    def forward(ctx, zr, zi, tau):
        require_device(zr, zi)
        ctx.fmt = fmt = _layout_of(zr)
        zr, zi = _al16(_cf(zr, fmt)), _al16(_cf(zi, fmt))
        tp, tv, tn, keep = _tau_args(tau, zr, fmt)
        yr, yi = torch.empty_like(zr), torch.empty_like(zi)
        call("cplxamd_modrelu_fwd", ptr(zr), ptr(zi), ptr(tp), tv, tn, ptr(yr), ptr(yi), zr.numel(),
             dtype_code(zr), stream_ptr())
        ctx.save_for_backward(zr, zi, *([keep] if keep is not None else []))
        ctx.tau = (tv, tn, tau.shape if isinstance(tau, torch.Tensor) else None)
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        zr, zi, *rest = ctx.saved_tensors
        tv, tn, tshape = ctx.tau
        tp = rest[0] if rest else None
        gr, gi = _al16(_cf(gr, ctx.fmt)), _al16(_cf(gi, ctx.fmt))
        dzr, dzi = torch.empty_like(zr), torch.empty_like(zi)
        want_tau = tshape is not None and ctx.needs_input_grad[2]
        dtau = torch.empty(zr.shape, dtype=torch.float32, device=zr.device, memory_format=ctx.fmt) if want_tau else None
        call("cplxamd_modrelu_bwd", ptr(zr), ptr(zi), ptr(tp), tv, tn, ptr(gr), ptr(gi), ptr(dzr),
             ptr(dzi), ptr(dtau), zr.numel(), dtype_code(zr), stream_ptr())
        if want_tau:
            dtau = dtau.sum_to_size(tshape) if len(tshape) else dtau.sum()
        return dzr, dzi, dtau


def _cplx_mul(ar, ai, br, bi, div=False, conj_b=False, neg=False):
    outr, outi = torch.empty_like(ar), torch.empty_like(ai)
    call("cplxamd_cplx_mul", ptr(ar), ptr(ai), ptr(br), ptr(bi), ptr(outr), ptr(outi), ar.numel(), int(div), int(conj_b),
         int(neg), dtype_code(ar), stream_ptr())
    return outr, outi


def cplx_mul_ok(ar, ai, br, bi):
    """Both operands planar complex device tensors of one shape / dtype (float32 or bf16): the one-launch kernel applies
    (broadcasting products, scalars and CPU tensors keep torch's elementwise kernels)."""
    return (ar.is_cuda and ar.dtype in (torch.float32, torch.bfloat16) and ar.numel() > 0
            and all(t.shape == ar.shape and t.dtype == ar.dtype and t.device == ar.device for t in (ai, br, bi)))


class CplxMulFn(torch.autograd.Function):
    """Cplx * Cplx and Cplx / Cplx (cplxmodule/cplx.py:135-165) in one launch (the reference: 6 / 12 elementwise kernels,
    same operation order -> same bits for float32; bf16 planes keep float32 intermediates where the reference's chain of
    torch ops rounds every product and sum to bf16: more accurate, not bit-identical); gradients d(ab)/da = g conj(b), d(ab)/db = g conj(a), d(a/b)/da = g / conj(b),
    d(a/b)/db = -(g conj(a/b)) / conj(b): one launch each (two for the last)."""

    @staticmethod
    def forward(ctx, ar, ai, br, bi, div):
        require_device(ar, ai, br, bi)
        ctx.fmt = fmt = _layout_of(ar)
        # what is saved are the tensors AS GIVEN: the dense / aligned copies made here have no autograd history, and a
        # create_graph backward built on them would silently drop the second-order terms of a transposed, channels-last or
        # misaligned operand; the raw-kernel backward redoes the (usually free) conversion
        raw = (ar, ai, br, bi)
        ar, ai, br, bi = (_al16(_cf(t, fmt)) for t in raw)
        yr, yi = _cplx_mul(ar, ai, br, bi, div=div)
        ctx.div = div
        ctx.save_for_backward(*((raw[2], raw[3], yr, yi) if div else raw))
        return yr, yi

    @staticmethod
    def backward(ctx, gr, gi):
        need_a, need_b = ctx.needs_input_grad[0] or ctx.needs_input_grad[1], ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        da = db = (None, None)
        if torch.is_grad_enabled():
            # create_graph=True (gradient penalties, Hessian-vector products): the reference's product is a composition of
            # differentiable torch ops (cplx.py:135-146), so its backward is differentiable too -- the raw kernels below are
            # not.  Same formulas, spelled with torch ops on the saved (graph-connected) tensors.
            gr = torch.zeros_like(ctx.saved_tensors[0]) if gr is None else gr
            gi = torch.zeros_like(ctx.saved_tensors[0]) if gi is None else gi
            if ctx.div:
                br, bi, yr, yi = ctx.saved_tensors
                n2 = br * br + bi * bi
                if need_a:            # g / conj(b) = g b / |b|^2
                    da = ((gr * br - gi * bi) / n2, (gr * bi + gi * br) / n2)
                if need_b:            # -g conj(y) / conj(b)
                    tr, ti = gr * yr + gi * yi, gi * yr - gr * yi
                    db = (-(tr * br - ti * bi) / n2, -(tr * bi + ti * br) / n2)
            else:
                ar, ai, br, bi = ctx.saved_tensors
                if need_a:            # g conj(b)
                    da = (gr * br + gi * bi, gi * br - gr * bi)
                if need_b:            # g conj(a)
                    db = (gr * ar + gi * ai, gi * ar - gr * ai)
            return da[0], da[1], db[0], db[1], None
        gr, gi = _al16(_cf(gr, ctx.fmt)), _al16(_cf(gi, ctx.fmt))
        dense = lambda *ts: tuple(_al16(_cf(t, ctx.fmt)) for t in ts)  # noqa: E731
        if ctx.div:
            br, bi, yr, yi = ctx.saved_tensors
            br, bi = dense(br, bi)
            if need_a:
                da = _cplx_mul(gr, gi, br, bi, div=True, conj_b=True)
            if need_b:
                t = _cplx_mul(gr, gi, yr, yi, conj_b=True)
                db = _cplx_mul(t[0], t[1], br, bi, div=True, conj_b=True, neg=True)
        else:
            ar, ai, br, bi = dense(*ctx.saved_tensors)
            if need_a:
                da = _cplx_mul(gr, gi, br, bi, conj_b=True)
            if need_b:
                db = _cplx_mul(gr, gi, ar, ai, conj_b=True)
        return da[0], da[1], db[0], db[1], None


def cplx_mul(ar, ai, br, bi, div=False):
    return CplxMulFn.apply(ar, ai, br, bi, bool(div))


class SplitReluFn(torch.autograd.Function):
    """torch.nn.ReLU on both planes (CplxToCplx[torch.nn.ReLU], cplxmodule/nn/modules/base.py:167-199): one launch
    forward, one backward (the mask is read off the saved outputs, as aten's threshold_backward does)."""

    @staticmethod
    def forward(ctx, xr, xi):
        require_device(xr, xi)
        ctx.fmt = fmt = _layout_of(xr)
        xr, xi = _al16(_cf(xr, fmt)), _al16(_cf(xi, fmt))
        yr, yi = torch.empty_like(xr), torch.empty_like(xi)
        call("cplxamd_split_relu", ptr(xr), ptr(xi), None, None, ptr(yr), ptr(yi), xr.numel(), 0, dtype_code(xr), stream_ptr())
        ctx.save_for_backward(yr, yi)
        return yr, yi

    @staticmethod
    def backward(ctx, gr, gi):
        yr, yi = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True: differentiable spelling (d/dg of g * [y > 0]; see CplxMulFn.backward)
            return (None if gr is None else gr * (yr > 0).to(gr.dtype)), (None if gi is None else gi * (yi > 0).to(gi.dtype))
        gr, gi = _al16(_cf(gr, ctx.fmt)), _al16(_cf(gi, ctx.fmt))
        dxr, dxi = torch.empty_like(yr), torch.empty_like(yi)
        call("cplxamd_split_relu", ptr(yr), ptr(yi), ptr(gr), ptr(gi), ptr(dxr), ptr(dxi), yr.numel(), 1, dtype_code(yr),
             stream_ptr())
        return dxr, dxi


def split_relu(xr, xi):
    return SplitReluFn.apply(xr, xi)


class CplxDropoutFn(torch.autograd.Function):
    """One keep / drop decision per complex element (nn/modules/extra.py:7-25); the mask is a
    function of (seed, offset) and is regenerated in backward."""

    @staticmethod
    def forward(ctx, xr, xi, p, seed, offset):
        require_device(xr, xi)
        ctx.fmt = fmt = _layout_of(xr)       # (the mask is a function of the position in STORAGE order: same layout in backward)
        xr, xi = _al16(_cf(xr, fmt)), _al16(_cf(xi, fmt))
        ctx.p, ctx.seed, ctx.offset = p, seed, offset
        return CplxDropoutFn._run(xr, xi, p, seed, offset)

    @staticmethod
    def _run(xr, xi, p, seed, offset):
        yr, yi = torch.empty_like(xr), torch.empty_like(xi)
        sd, of, st = _noise_args(seed, offset)
        call("cplxamd_cplx_dropout", ptr(xr), ptr(xi), ptr(yr), ptr(yi), float(p), sd, of, st, xr.numel(),
             dtype_code(xr), stream_ptr())
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        dr, di = CplxDropoutFn._run(_al16(_cf(gr, ctx.fmt)), _al16(_cf(gi, ctx.fmt)), ctx.p, ctx.seed, ctx.offset)
        return dr, di, None, None, None


def _pool_out(L, k, s, p, d, ceil_mode):
    """torch's pooling output size (incl. the ceil_mode rule that the last window must start
    inside the input or its left padding)."""
    num = L + 2 * p - d * (k - 1) - 1
    o = (-(-num // s) if ceil_mode else num // s) + 1
    if ceil_mode and (o - 1) * s >= L + p:
        o -= 1
    return o


class CplxMaxPool2dFn(torch.autograd.Function):
    """cplx.max_pool2d (cplxmodule/cplx.py:1114-1175, 1183-1190) as one kernel + gather backward."""

    @staticmethod
    def forward(ctx, zr, zi, kernel, stride, padding, dilation, ceil_mode):
        import ctypes
        require_device(zr, zi)
        ctx.fmt = fmt = _layout_of(zr)       # channels-last in -> channels-last out (and index map)
        zr, zi = _cf(zr, fmt), _cf(zi, fmt)
        B, C, H, W = zr.shape
        (kh, kw), (sh, sw), (ph, pw), (dh, dw) = kernel, stride, padding, dilation
        if ph > kh // 2 or pw > kw // 2:
            raise RuntimeError("pad should be at most half of effective kernel size")
        Ho, Wo = _pool_out(H, kh, sh, ph, dh, ceil_mode), _pool_out(W, kw, sw, pw, dw, ceil_mode)
        pool = (ctypes.c_int * 14)(B, C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw)
        yr = torch.empty((B, C, Ho, Wo), dtype=zr.dtype, device=zr.device, memory_format=fmt)
        yi = torch.empty_like(yr)
        idx = torch.empty((B, C, Ho, Wo), dtype=torch.int32, device=zr.device, memory_format=fmt)
        ctx.sfx = sfx = "_cl" if fmt == torch.channels_last else ""
        call("cplxamd_cplx_maxpool2d_fwd" + sfx, ptr(zr), ptr(zi), ptr(yr), ptr(yi), ptr(idx), pool,
             dtype_code(zr), stream_ptr())
        ctx.save_for_backward(idx)
        ctx.pool, ctx.in_shape = pool, zr.shape
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        (idx,) = ctx.saved_tensors
        gr, gi = _cf(gr, ctx.fmt), _cf(gi, ctx.fmt)
        dzr = torch.empty(ctx.in_shape, dtype=gr.dtype, device=gr.device, memory_format=ctx.fmt)
        dzi = torch.empty_like(dzr)
        call("cplxamd_cplx_maxpool2d_bwd" + ctx.sfx, ptr(gr), ptr(gi), ptr(idx), ptr(dzr), ptr(dzi), ctx.pool,
             dtype_code(gr), stream_ptr())
        return dzr, dzi, None, None, None, None, None


# ------------------------------------------------------------------------------------------ #
#  float64: a parity mode on its own kernels (cplxmodule_amd/f64.py, csrc/f64.hip)           #
# ------------------------------------------------------------------------------------------ #
class Route:
    """`.apply` of an autograd Function that hands float64 arguments (the dtype of its first tensor argument) to the
    float64 implementation in f64.py and everything else to the Function.  Every float32 / bf16 call pays one dtype test."""

    def __init__(self, fn, f64_name):
        self.fn, self.f64_name = fn, f64_name
        self.__name__, self.__doc__ = fn.__name__, fn.__doc__

    def apply(self, *args):
        for a in args:
            if isinstance(a, torch.Tensor):
                if a.dtype == torch.float64:
                    from . import f64
                    return getattr(f64, self.f64_name)(*args)
                break
        return self.fn.apply(*args)

    def __getattr__(self, name):            # (class attributes / static helpers of the Function)
        return getattr(self.fn, name)


CplxLinearFn = Route(CplxLinearFn, "cplx_linear")
RealLinearFn = Route(RealLinearFn, "real_linear")
CplxLinearLRTFn = Route(CplxLinearLRTFn, "cplx_linear_lrt")
RealLinearLRTFn = Route(RealLinearLRTFn, "real_linear_lrt")
LogAlphaFn = Route(LogAlphaFn, "log_alpha")
PenaltyFn = Route(PenaltyFn, "penalty")
PenaltySumFn = Route(PenaltySumFn, "penalty_sum")
AbsFn = Route(AbsFn, "cplx_abs")
ExpiFn = Route(ExpiFn, "expi")
MaskMulFn = Route(MaskMulFn, "mask_mul")
_relevance_mask32 = relevance_mask


def relevance_mask(wr, wi, ls2, threshold, count=False):  # noqa: F811
    if ls2.dtype == torch.float64:
        from . import f64
        return f64.relevance_mask(wr, wi, ls2, threshold, count)
    return _relevance_mask32(wr, wi, ls2, threshold, count)


relevance_mask.__doc__ = _relevance_mask32.__doc__
