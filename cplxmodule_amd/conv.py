"""Complex / real 2-d convolution on the implicit-GEMM kernels (csrc/conv.hip), with autograd,
and the conv flavours of the local-reparameterization layers.

Reference: cplx.conv2d -> convnd (cplxmodule/cplx.py:770-838), CplxConvNdGaussianMixin
(nn/relevance/complex/base.py:120-135), ConvNdGaussianMixin (nn/relevance/real/base.py:116-163).
"""
import ctypes
import os
import weakref

import torch
import torch.nn.functional as F

from . import _lib, ops
from .ops import once_differentiable
from ._lib import CplxAmdError, call, try_call, dtype_code, launch_flags, ptr, require_device, stream_ptr
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
    key = _lib.scratch_key(device)
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


def nhwc_pad(x, ph, pw, Hp=None, Wp=None, head_rows=0, tail_rows=0):
    """planar NCHW bf16 -> zero-padded channels-last grid [B, Hp, Wp, C] (default: symmetric
    padding), inside a buffer with `head_rows` / `tail_rows` extra zero rows of C before / after
    the grid (the kernels index rows relative to the grid and never clamp inside those)."""
    B, C, H, W = x.shape
    Hp = H + 2 * ph if Hp is None else Hp
    Wp = W + 2 * pw if Wp is None else Wp
    rows = B * Hp * Wp
    buf = torch.empty(head_rows + rows + tail_rows, C, dtype=x.dtype, device=x.device)
    grid = buf[head_rows:head_rows + rows]
    entry = "cplxamd_nhwc_pad" if x.dtype == torch.bfloat16 else "cplxamd_nhwc_pad_f32"
    call(entry, ptr(x), ptr(grid), B, C, H, W, ph, pw, Hp, Wp, stream_ptr())
    if head_rows:
        buf[:head_rows].zero_()
    if tail_rows:
        buf[head_rows + rows:].zero_()
    return grid.view(B, Hp, Wp, C)


# ---- eligibility of the channels-last ("rows") kernels: stride 1, groups 1 -------------------- #
_ROWS_MIN_FLOP = 4e9
_ROWS_FORCE = False          # tests set this to exercise the kernels on tiny shapes
def _grid(geom):
    return geom[3] + 2 * geom[9], geom[4] + 2 * geom[10]             # Hp, Wp of the input grid


def _rows_base_ok(geom):
    """stride 1, groups 1, a halo of at most 32 columns -- and enough work to pay for the
    channels-last copies and the 64-channel tiles: small layers (a few GFLOP, < 32 channels) are
    launch-bound and stay on the single-launch gather kernels."""
    Hp, Wp = _grid(geom)
    flop = 8.0 * geom[0] * Hp * Wp * geom[1] * geom[2] * geom[5] * geom[6]
    return (geom[7] == 1 and geom[8] == 1 and geom[13] == 1 and (geom[6] - 1) * geom[12] <= 32
            and geom[0] * Hp * Wp < 2 ** 31 and (flop >= _ROWS_MIN_FLOP and min(geom[1], geom[2]) >= 32
                                                or _ROWS_FORCE))


def _rows_fwd_ok(geom, dtype=torch.bfloat16):
    """bf16 stages hold 32 channels, float32 stages 16."""
    return _rows_base_ok(geom) and geom[1] % (32 if dtype == torch.bfloat16 else 16) == 0


def _rows_dgrad_ok(geom, dtype=torch.bfloat16):
    return _rows_base_ok(geom) and geom[2] % (32 if dtype == torch.bfloat16 else 16) == 0


def _rows_wgrad_ok(geom, cplx, dtype=torch.bfloat16):
    Hp, Wp = _grid(geom)
    KH, KW = geom[5], geom[6]
    bf16 = dtype == torch.bfloat16
    span = geom[0] * Hp * Wp + 64 + KH * geom[11] * Wp
    cmul = 8 if bf16 else 4
    if not (_rows_base_ok(geom) and geom[1] % cmul == 0 and geom[2] % cmul == 0
            and span * max(geom[1], geom[2]) < 2 ** 31):
        return False
    if KH == 1 and KW == 1:
        return True                      # 1x1: the (T, T) GEMM of the linear layer on the two grids
    # KW waves stage kr + kr + (KW-1)*dil_w rows (kr = 32 bf16 / 16 float32 rows of 8 / 16 chunks)
    # per plane in at most 8 pieces each
    kr, per_row = (32, 8) if bf16 else (16, 16)
    chunks = (2 if cplx else 1) * per_row * (2 * kr + (KW - 1) * geom[12])
    return KW <= 4 and -(-chunks // (64 * KW)) <= 8


def _shift_rows(geom):
    """Largest tap shift in grid rows: (KH-1)*dil_h*Wp + (KW-1)*dil_w."""
    return (geom[5] - 1) * geom[11] * _grid(geom)[1] + (geom[6] - 1) * geom[12]


def input_grid(xr, xi, geom):
    """Channels-last copy of the (padded) input: read by the forward and by the weight gradient."""
    tail = 320 + _shift_rows(geom)       # the forward kernel stages 320-row windows without clamping
    return (nhwc_pad(xr, geom[9], geom[10], tail_rows=tail),
            None if xi is None else nhwc_pad(xi, geom[9], geom[10], tail_rows=tail))


def grad_grid(gr, gi, geom):
    """The output gradient laid top-left on the input's padded grid: read by the data gradient
    (backwards, hence the zero head rows) and by the weight gradient (zero tail up to 32 rows)."""
    Hp, Wp = _grid(geom)
    head, tail = _shift_rows(geom), 320 + (-(geom[0] * Hp * Wp)) % 32
    return (nhwc_pad(gr, 0, 0, Hp, Wp, head_rows=head, tail_rows=tail),
            None if gi is None else nhwc_pad(gi, 0, 0, Hp, Wp, head_rows=head, tail_rows=tail))


# ---- channels-last end to end (csrc/conv_cl.hip): 3 x 3-style "same" convolutions, stride 1, groups 1 -------- #
_CL_FORCE = False            # tests: take the kernels on tiny shapes too
_CL_ENABLED = True
_CL_PATCH = os.environ.get("CPLXAMD_CL_PATCH", "1") != "0"      # conv_cl2.hip where it applies (A/B: set to 0)
_CL_MIN_FLOP = 4e9


def _cl_ok(geom, dgrad=False):
    """Can the persistent channels-last kernel run this convolution (forward) / its data gradient?  Zero padding up to
    `same`, KW = 3, contraction channels a multiple of 16 with KH * C/16 a multiple of 6 (the ring is unrolled over
    3 slots x 2 fragment register sets), output channels a multiple of 64, one activation plane below ~3.7 GiB."""
    B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups = (geom[i] for i in range(14))
    C, N = (Co, Ci) if dgrad else (Ci, Co)
    if not _CL_ENABLED or sh != 1 or sw != 1 or groups != 1 or KW != 3 or KH > 8:
        return False
    if 2 * ph > dh * (KH - 1) or 2 * pw > dw * (KW - 1) or (KW - 1) * dw > 64:
        return False
    if C % 16 or N % 64 or (KH * (C // 16)) % 6:
        return False
    P = B * H * W
    if P == 0 or P >= 2 ** 31 - 512 or P * C * 2 + 4 * (dh * (KH - 1) * W + dw * (KW - 1) + 512) * C >= 0xF0000000:
        return False
    return _CL_FORCE or 8.0 * P * Ci * Co * KH * KW >= _CL_MIN_FLOP


def _cl_wgrad_ok(geom):
    """3 x 3 convolution with padding up to `same`, channel counts multiples of 64 (any image width: a row's last
    32-pixel stage may be short)."""
    B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups = (geom[i] for i in range(14))
    if not _CL_ENABLED or sh != 1 or sw != 1 or groups != 1 or KH != 3 or KW != 3 or ph > dh or pw > dw or dw > 4:
        return False
    if H + 2 * ph - 2 * dh <= 0 or W + 2 * pw - 2 * dw <= 0:
        return False
    P = B * H * W
    if Ci % 64 or Co % 64 or P == 0 or P >= 2 ** 31 or (P + ph * W + 64) * max(Ci, Co) * 2 >= 2 ** 32 - 64:
        return False
    return _CL_FORCE or 8.0 * P * Ci * Co * 9 >= _CL_MIN_FLOP


def cl_wgrad(gr, gi, xr, xi, geom, w_shape, emul=None):
    """dW (float32, [Co, Ci, 3, 3] planes) from channels-last gradient / input planes (csrc/conv_cl_wgrad.hip)."""
    B, Ci, Co, H, W = (geom[i] for i in range(5))
    gr, gi, xr, xi = (to_channels_last(t) for t in (gr, gi, xr, xi))
    ws = _scratch(gr.device, int(_lib.load().cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)))
    dwr = torch.empty(w_shape, dtype=torch.float32, device=gr.device)
    dwi = torch.empty_like(dwr)
    call("cplxamd_conv2d_cl_wgrad_fl", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(emul), ptr(dwr), ptr(dwi), B, H, W, Ci, Co,
         geom[5], geom[6], geom[11], geom[12], geom[9], geom[10], ptr(ws), ws.numel(), launch_flags(), stream_ptr())
    return dwr, dwi


# The batch-norm backward folded into the weight gradient (csrc/conv_cl_wgrad.hip FOLD; A/B: CPLXAMD_BN_FOLD=0).
_BN_FOLD = os.environ.get("CPLXAMD_BN_FOLD", "1") != "0"


def bn_fold_node(xr, xi):
    """Called by the batch-norm forward with its input planes: the autograd node of the channels-last 3 x 3 convolution
    that produced exactly these two tensors (and whose weight gradient the folded kernel can form), or None."""
    node = xr.grad_fn
    if (not _BN_FOLD or node is None or node is not xi.grad_fn or not isinstance(node, CplxConv2dFn.fn._backward_cls)
            or xr.output_nr != 0 or xi.output_nr != 1 or not getattr(node, "cl", False)):
        return None
    if xr.dtype != torch.bfloat16 or not _cl_wgrad_ok(node.geom) or node.wshape[0] % 64:
        return None
    return node


def cl_wgrad_bn(gr, gi, zr, zi, coef, node):
    """-> (dy_r, dy_i, (dW_r, dW_i, node)) or None: the input gradient of the batch-norm layer (coef: cplxamd_bn_bwd_coef)
    formed inside the weight-gradient launch of the convolution `node` that produced z."""
    if not (node.needs_input_grad[2] or node.needs_input_grad[3]):
        return None
    xr, xi = node.saved_tensors[:2]
    geom = node.geom
    B, Ci, Co, H, W = (geom[i] for i in range(5))
    if tuple(zr.shape) != tuple(gr.shape) or zr.shape[1] != Co:
        return None
    ws = _scratch(gr.device, int(_lib.load().cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)))
    dyr, dyi = torch.empty_like(zr), torch.empty_like(zi)
    dwr = torch.empty(node.wshape, dtype=torch.float32, device=gr.device)
    dwi = torch.empty_like(dwr)
    if not try_call("cplxamd_conv2d_cl_wgrad_bn_fl", ptr(gr), ptr(gi), ptr(zr), ptr(zi), ptr(coef), ptr(xr), ptr(xi), ptr(dyr),
                    ptr(dyi), ptr(dwr), ptr(dwi), B, H, W, Ci, Co, geom[5], geom[6], geom[11], geom[12], geom[9], geom[10],
                    ptr(ws), ws.numel(), launch_flags(), stream_ptr()):
        return None
    return dyr, dyi, (dwr, dwi, node)


def to_channels_last(t):
    """[B, C, H, W] -> the same logical tensor stored [B, H, W, C] (torch.channels_last); no copy if it already is."""
    if t.is_contiguous(memory_format=torch.channels_last):
        return t
    B, C, H, W = t.shape
    if t.is_contiguous() and t.dtype == torch.bfloat16 and C % 8 == 0 and B * (-(-C // 32)) <= 65535 and H <= 65535 \
            and t.data_ptr() % 16 == 0:
        return nhwc_pad(t, 0, 0).permute(0, 3, 1, 2)
    return t.contiguous(memory_format=torch.channels_last)


def from_channels_last(t):
    """A channels-last [B, C, H, W] tensor -> plain contiguous (NCHW) storage, by the LDS-tile transpose when it can."""
    if t.is_contiguous():
        return t
    B, C, H, W = t.shape
    if (t.is_contiguous(memory_format=torch.channels_last) and t.dtype == torch.bfloat16 and C % 8 == 0 and (H * W) % 8 == 0
            and B * (-(-C // 64)) <= 65535 and t.data_ptr() % 16 == 0):
        out = torch.empty((B, C, H, W), dtype=t.dtype, device=t.device)
        call("cplxamd_cl_to_nchw", ptr(t), ptr(out), B, C, H * W, stream_ptr())
        return out
    return t.contiguous()


class ToChannelsLastFn(torch.autograd.Function):
    """Differentiable to_channels_last for callers outside the conv Functions: forward and backward are the LDS-tile
    transposes (the gradient goes back in the layout the input came in)."""

    @staticmethod
    def forward(ctx, t):
        ctx.planar = t.is_contiguous() and not t.is_contiguous(memory_format=torch.channels_last)
        return to_channels_last(t)

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():        # create_graph: a layout change is a differentiable torch op
            return g.contiguous() if ctx.planar else g
        return from_channels_last(g) if ctx.planar else g


def _cl_pack(wr, wi, dgrad):
    """bf16 weight planes [Co, Ci, KH, KW] -> the per-stage LDS images of conv_cl.hip."""
    Co, Ci, KH, KW = wr.shape
    N, C = (Ci, Co) if dgrad else (Co, Ci)
    nbytes = int(_lib.load().cplxamd_conv2d_cl_pack_bytes(N, C, KH, KW))
    out = torch.empty(nbytes, dtype=torch.uint8, device=wr.device)
    call("cplxamd_conv2d_cl_pack", ptr(wr), ptr(wi), ptr(out), Co, Ci, KH, KW, int(dgrad), stream_ptr())
    return out


# Convolutions whose output a training-mode batch-norm layer consumed recently (keyed by the weight parameter, weakly):
# their next forward also forms that layer's statistics in its epilogue (cplxamd_conv2d_cl2_mom) -- armed by the
# consumer, like the KL fusion of the relevance layers, so a convolution nothing normalises never pays for it.
#  * A request is a CREDIT, not a switch: every armed forward spends one, every consuming batch-norm forward refills it to
#    _MOMENTS_CREDIT.  A convolution whose output stops feeding a batch-norm layer (the model was edited, the layer went to
#    evaluation mode, another consumer took over) therefore pays the +10 % epilogue for at most that many more steps.
#  * Implicit arming means step 1 (the layer's own moment pass: float64 sums over the stored output) and steps >= 2 (the
#    epilogue's per-workgroup float32 partial sums, summed in float64) take different kernels; both are within 1e-6 of the
#    exact moments but not bit-equal to each other, and the epilogue's summation order follows the launch grid (i.e. the
#    per-call launch flags).  For bit-reproducible statistics from the first step on, arm at model-build time with
#    `arm_conv_bn(model)` (permanent requests), or switch the path off (CPLXAMD_CONV_BN_MOMENTS=0).
_MOMENTS_WANTED = {}          # id(weight plane) -> [weak reference to it, credit]   (tensors compare elementwise: no WeakSet)
_MOMENTS = os.environ.get("CPLXAMD_CONV_BN_MOMENTS", "1") != "0"      # (A/B: 0 = the layer's own moment pass, always)
_MOMENTS_CREDIT = 2
_PERMANENT = 1 << 60


def want_moments(weight_plane, on=True, permanent=False):
    """Called by the batch-norm forward with the tag the convolution left on its output (`_cplxamd_conv_src`);
    on=False withdraws the request (evaluation mode, synchronised statistics) unless it is a permanent one."""
    if weight_plane is None:
        return
    key = id(weight_plane)
    cur = _MOMENTS_WANTED.get(key)
    if cur is not None and cur[0]() is not weight_plane:
        cur = None                                     # (an id re-used by another tensor)
    if on:
        credit = _PERMANENT if permanent or (cur is not None and cur[1] >= _PERMANENT) else _MOMENTS_CREDIT
        if cur is None:
            _MOMENTS_WANTED[key] = [weakref.ref(weight_plane, lambda _, k=key: _MOMENTS_WANTED.pop(k, None)), credit]
        else:
            cur[1] = credit
    elif cur is None or cur[1] < _PERMANENT or permanent:
        _MOMENTS_WANTED.pop(key, None)


def moments_wanted(weight_plane, spend=False):
    """Is the moments epilogue requested for this convolution?  spend=True (the convolution's forward): uses up one
    credit of the request."""
    r = _MOMENTS_WANTED.get(id(weight_plane))
    if r is None or r[0]() is not weight_plane:
        return False
    if spend and r[1] < _PERMANENT:
        r[1] -= 1
        if r[1] <= 0:
            _MOMENTS_WANTED.pop(id(weight_plane), None)
    return True


def arm_conv_bn(module, on=True):
    """Arm (permanently, at model-build time) the moments epilogue of every CplxConv2d that is DIRECTLY followed by a
    CplxBatchNorm2d inside a torch.nn.Sequential of `module`: the first training step then runs the same kernels as every
    later one (bit-reproducible batch statistics from step 1, a hipGraph captured without warm-up steps captures the armed
    variant).  The pair still falls back to the layer's own moment pass whenever the kernel variant does not take the
    shape or the launch flags (float32, dilation, CPLXAMD_LAUNCH_SHARED).  Returns the number of pairs (un)armed."""
    from .nn.modules.batchnorm import CplxBatchNorm2d
    from .nn.modules.conv import CplxConv2d
    n = 0
    for m in module.modules():
        if isinstance(m, torch.nn.Sequential):
            kids = list(m.children())
            for a, b in zip(kids, kids[1:]):
                if type(a) is CplxConv2d and isinstance(b, CplxBatchNorm2d):
                    want_moments(a.weight.real, on=on, permanent=True)
                    n += 1
    return n


def cl_conv(xr, xi, wr, wi, br, bi, geom, dgrad=False, moments=False):
    """Forward (or, with dgrad, the data gradient read as a convolution of the output gradient with the flipped,
    conjugated, channel-swapped kernel) on channels-last planes; returns channels-last [B, N, H, W] tensors.
    moments (forward only): also leave the batch-norm moments of the output on it (ops.attach_moments) when the
    kernel variant takes the shape."""
    B, Ci, Co, H, W, KH, KW = (geom[i] for i in range(7))
    C, N = (Co, Ci) if dgrad else (Ci, Co)
    xr, xi = to_channels_last(xr), to_channels_last(xi)
    wp = _cl_pack(wr, wi, dgrad)
    Ho = H + 2 * geom[9] - geom[11] * (KH - 1)
    Wo = W + 2 * geom[10] - geom[12] * (KW - 1)
    oshape = (B, N, H, W) if dgrad else (B, N, Ho, Wo)
    yr = torch.empty(oshape, dtype=xr.dtype, device=xr.device, memory_format=torch.channels_last)
    yi = torch.empty_like(yr)
    ws = _scratch(xr.device, int(_lib.load().cplxamd_conv2d_cl_ws_bytes(N)))
    flags = launch_flags()          # one policy for the whole call (the moments variant and its chunk count must agree)
    args = (ptr(xr), ptr(xi), ptr(wp), ptr(br), ptr(bi), ptr(yr), ptr(yi), B, H, W, C, N, KH, KW, geom[11], geom[12], geom[9],
            geom[10], int(dgrad), ptr(ws), ws.numel(), flags, stream_ptr())
    if moments and not dgrad and _CL_PATCH and _MOMENTS:
        chunks = int(_lib.load().cplxamd_conv2d_cl2_mom_chunks_fl(B, H, W, C, N, KH, KW, geom[11], geom[12], geom[9], geom[10],
                                                                  flags))
        if chunks > 0:
            partials = torch.empty(chunks * N * 5, dtype=torch.float64, device=xr.device)
            if try_call("cplxamd_conv2d_cl2_mom_fl", *args[:18], ptr(partials), partials.numel() * 8, ptr(ws), ws.numel(),
                        flags, stream_ptr()):
                ops.attach_moments(yr, yi, partials, chunks)
                return yr, yi
    # dilation 1: the 2-d-patch kernel (activations staged once per channel slice for all nine taps); else the row kernel
    if not (_CL_PATCH and try_call("cplxamd_conv2d_cl2_fl", *args)):
        call("cplxamd_conv2d_cl_fl", *args)
    return yr, yi


_LRT_DX_FUSE = os.environ.get("CPLXAMD_LRT_DX_FUSE", "1") != "0"      # (A/B: set to 0 for the two launches)


def cl_conv_lrt_dx(gr, gi, wr, wi, geom, xr, xi, ga):
    """Input gradient of a local-reparameterization convolution on channels-last planes,
    dx = dgrad(g; w) + 2 x (*) ga: one launch (cplxamd_conv2d_cl2_lrt_dx, the sum formed in the data-gradient kernel's
    epilogue) where the 2-d-patch kernel applies, else the data gradient followed by `ops.lrt_dx_accum` (same bits)."""
    B, Ci, Co, H, W, KH, KW = (geom[i] for i in range(7))
    if _LRT_DX_FUSE and _CL_PATCH and ga.dtype == xr.dtype == torch.bfloat16 and KH == KW == 3 and geom[11] == geom[12] == 1:
        gr, gi, ga = to_channels_last(gr), to_channels_last(gi), to_channels_last(ga)
        xr, xi = to_channels_last(xr), to_channels_last(xi)
        wp = _cl_pack(wr, wi, True)
        dxr = torch.empty((B, Ci, H, W), dtype=xr.dtype, device=xr.device, memory_format=torch.channels_last)
        dxi = torch.empty_like(dxr)
        ws = _scratch(xr.device, int(_lib.load().cplxamd_conv2d_cl_ws_bytes(Ci)))
        if try_call("cplxamd_conv2d_cl2_lrt_dx_fl", ptr(gr), ptr(gi), ptr(wp), ptr(xr), ptr(xi), ptr(ga), ptr(dxr), ptr(dxi),
                    B, H, W, Co, Ci, geom[9], geom[10], ptr(ws), ws.numel(), launch_flags(), stream_ptr()):
            return dxr, dxi
    dxr, dxi = cl_conv(gr, gi, wr, wi, None, None, geom, dgrad=True)
    ops.lrt_dx_accum(dxr, dxi, xr, xi, ga)
    return dxr, dxi


def cl_conv_real(x, w, b, geom, dgrad=False):
    """Real-valued cl_conv (csrc/conv_cl_real.hip): the variance path of the LRT layers and the real VD / ARD layers."""
    B, Ci, Co, H, W, KH, KW = (geom[i] for i in range(7))
    C, N = (Co, Ci) if dgrad else (Ci, Co)
    x = to_channels_last(x)
    lib = _lib.load()
    wp = torch.empty(int(lib.cplxamd_conv2d_clr_pack_bytes(N, C, KH, KW)), dtype=torch.uint8, device=x.device)
    call("cplxamd_conv2d_clr_pack", ptr(w), ptr(wp), Co, Ci, KH, KW, int(dgrad), stream_ptr())
    Ho = H + 2 * geom[9] - geom[11] * (KH - 1)
    Wo = W + 2 * geom[10] - geom[12] * (KW - 1)
    oshape = (B, N, H, W) if dgrad else (B, N, Ho, Wo)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws = _scratch(x.device, int(lib.cplxamd_conv2d_clr_ws_bytes(N)))
    call("cplxamd_conv2d_clr_fl", ptr(x), ptr(wp), ptr(b), ptr(y), B, H, W, C, N, KH, KW, geom[11], geom[12], geom[9],
         geom[10], int(dgrad), ptr(ws), ws.numel(), launch_flags(), stream_ptr())
    return y


def cl_wgrad_real(g, x, geom, w_shape, emul=None, emul_exp=False):
    """dW (float32 [Co, Ci, 3, 3]) = (sum g x) * emul (or * exp(emul)) from channels-last planes."""
    B, Ci, Co, H, W = (geom[i] for i in range(5))
    g, x = to_channels_last(g), to_channels_last(x)
    ws = _scratch(g.device, int(_lib.load().cplxamd_conv2d_clr_wgrad_ws_bytes(B, H, W, Ci, Co)))
    dw = torch.empty(w_shape, dtype=torch.float32, device=g.device)
    call("cplxamd_conv2d_clr_wgrad_fl", ptr(g), ptr(x), ptr(emul), int(emul_exp), ptr(dw), B, H, W, Ci, Co, geom[5], geom[6],
         geom[11], geom[12], geom[9], geom[10], ptr(ws), ws.numel(), launch_flags(), stream_ptr())
    return dw


def _cl_layer_ok(geom, *tensors):
    """A whole layer (forward, data gradient, weight gradient) on the channels-last kernels?"""
    return (all(t is None or t.dtype == torch.bfloat16 for t in tensors) and _cl_ok(geom) and _cl_ok(geom, dgrad=True)
            and _cl_wgrad_ok(geom))


def _pack_rows(w, swap):
    """[Co, Ci, KH, KW] -> [KH][KW][C/v][N][v] for the shifted-row kernels (v = 16 bf16 or 4 float32 =
    one MFMA operand load): (C, N) = (Ci, Co) for the forward, (Co, Ci) with both spatial dims
    flipped for the data gradient (`swap`)."""
    if swap:
        w = w.flip(2, 3).transpose(0, 1)
    N, C, KH, KW = w.shape
    v = 16 if w.dtype == torch.bfloat16 else 4
    return w.reshape(N, C // v, v, KH, KW).permute(3, 4, 1, 0, 2).contiguous()


def conv_fwd(xr, xi, wr, wi, br, bi, geom, out_shape, keep_grid=False):
    """-> (yr, yi) or, with keep_grid, (yr, yi, xp): xp = the channels-last input copies when the
    rows kernel ran (the caller keeps them for the weight gradient), else None."""
    yr = torch.empty(out_shape, dtype=xr.dtype, device=xr.device)
    yi = None if xi is None else torch.empty_like(yr)
    done = lambda xp: (yr, yi, xp) if keep_grid else (yr, yi)  # noqa: E731
    if geom[0] == 0:                       # empty batch: nothing to compute
        return done(None)
    if xr.dtype == torch.bfloat16 and _rows_fwd_ok(geom):
        xp = input_grid(xr, xi, geom)
        Hp, Wp = _grid(geom)
        wpr = _pack_rows(wr, False)
        wpi = None if wi is None else _pack_rows(wi, False)
        if try_call("cplxamd_conv2d_nhwc", ptr(xp[0]), ptr(xp[1]), ptr(wpr), ptr(wpi), ptr(br),
                    ptr(bi), ptr(yr), ptr(yi), geom[0], Hp, Wp, geom[1], geom[2], geom[5], geom[6],
                    geom[11], geom[12], 0, 0, 0, 0, out_shape[2], out_shape[3], dtype_code(yr),
                    stream_ptr()):
            return done(xp)
    if xr.dtype == torch.float32 and _rows_fwd_ok(geom, torch.float32):
        xp = input_grid(xr, xi, geom)
        Hp, Wp = _grid(geom)
        wpr = _pack_rows(wr, False)
        wpi = None if wi is None else _pack_rows(wi, False)
        if try_call("cplxamd_conv2d_nhwc_f32", ptr(xp[0]), ptr(xp[1]), ptr(wpr), ptr(wpi), ptr(br),
                    ptr(bi), ptr(yr), ptr(yi), geom[0], Hp, Wp, geom[1], geom[2], geom[5], geom[6],
                    geom[11], geom[12], 0, 0, 0, 0, out_shape[2], out_shape[3], stream_ptr()):
            return done(xp)
    if xr.dtype == torch.bfloat16 and try_call(
            "cplxamd_conv2d_bf16_fwd", ptr(xr), ptr(xi), ptr(wr), ptr(wi), ptr(br), ptr(bi), ptr(yr),
            ptr(yi), geom, ptr(_ktab(geom, 0, xr.device)), stream_ptr()):
        return done(None)
    call("cplxamd_conv2d_fwd", ptr(xr), ptr(xi), ptr(wr), ptr(wi), ptr(br), ptr(bi), ptr(yr),
         ptr(yi), geom, dtype_code(xr), stream_ptr())
    return done(None)


def conv_dgrad(gr, gi, wr, wi, geom, x_shape, gp=None):
    """gp: grad_grid(gr, gi, geom) if the caller already made it (shared with conv_wgrad)."""
    dxr = torch.empty(x_shape, dtype=gr.dtype, device=gr.device)
    dxi = None if gi is None else torch.empty_like(dxr)
    if geom[0] == 0:
        return dxr, dxi
    if gr.dtype == torch.float32 and _rows_dgrad_ok(geom, torch.float32):
        gp = grad_grid(gr, gi, geom) if gp is None else gp
        Hp, Wp = _grid(geom)
        wdr = _pack_rows(wr, True)
        wdi = None if wi is None else _pack_rows(wi, True)
        if try_call("cplxamd_conv2d_nhwc_f32", ptr(gp[0]), ptr(gp[1]), ptr(wdr), ptr(wdi), None, None,
                    ptr(dxr), ptr(dxi), geom[0], Hp, Wp, geom[2], geom[1], geom[5], geom[6],
                    geom[11], geom[12], 1, -_shift_rows(geom), geom[9], geom[10], x_shape[2],
                    x_shape[3], stream_ptr()):
            return dxr, dxi
    if gr.dtype == torch.bfloat16 and _rows_dgrad_ok(geom):
        # dX[h, w] sits at grid position (h + ph, w + pw) and reads the gradient grid backwards
        gp = grad_grid(gr, gi, geom) if gp is None else gp
        Hp, Wp = _grid(geom)
        wdr = _pack_rows(wr, True)
        wdi = None if wi is None else _pack_rows(wi, True)
        if try_call("cplxamd_conv2d_nhwc", ptr(gp[0]), ptr(gp[1]), ptr(wdr), ptr(wdi), None, None,
                    ptr(dxr), ptr(dxi), geom[0], Hp, Wp, geom[2], geom[1], geom[5], geom[6],
                    geom[11], geom[12], 1, -_shift_rows(geom), geom[9], geom[10], x_shape[2],
                    x_shape[3], dtype_code(dxr), stream_ptr()):
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


def _wgrad_rows(gp, xp, geom, w_shape, emul):
    """Weight gradient on the channels-last grids (csrc/conv_nhwc_wgrad.hip)."""
    B, Ci, Co, KH, KW = geom[0], geom[1], geom[2], geom[5], geom[6]
    cplx = gp[1] is not None
    Hp, Wp = _grid(geom)
    dev = gp[0].device
    if KH == 1 and KW == 1:
        # dW[co, ci] = sum_r G[r, co] conj(X[r, ci]): exactly the linear layer's weight gradient on
        # [rows, C] operands (both grids carry >= 32 zero tail rows, so K rounds up to 32 freely)
        K = -(-(B * Hp * Wp) // 32) * 32
        flat = lambda t, C: torch.as_strided(t, (K, C), (C, 1))  # noqa: E731
        if cplx:
            dwr, dwi = ops._cplx_linear_dw(flat(gp[0], Co), flat(gp[1], Co), flat(xp[0], Ci),
                                           flat(xp[1], Ci))
            return dwr.view(w_shape), dwi.view(w_shape)
        em = None if emul is None else emul.reshape(Co, Ci)
        return ops._real_linear_dw(flat(gp[0], Co), flat(xp[0], Ci), emul=em).view(w_shape), None
    sfx = "" if gp[0].dtype == torch.bfloat16 else "_f32"
    nbytes = int(getattr(_lib.load(), f"cplxamd_conv2d_nhwc_wgrad{sfx}_ws_bytes")(B, Hp, Wp, Ci, Co, KH, KW, int(cplx)))
    ws = _scratch(dev, nbytes)
    dwr = torch.empty(w_shape, dtype=torch.float32, device=dev)
    dwi = torch.empty_like(dwr) if cplx else None
    if try_call(f"cplxamd_conv2d_nhwc_wgrad{sfx}", ptr(gp[0]), ptr(gp[1]), ptr(xp[0]), ptr(xp[1]), ptr(emul),
                ptr(dwr), ptr(dwi), B, Hp, Wp, Ci, Co, KH, KW, geom[11], geom[12], ptr(ws),
                ws.numel(), stream_ptr()):
        return dwr, dwi
    return None


def conv_wgrad(gr, gi, xr, xi, geom, w_shape, emul=None, xp=None, gp=None, bias_out=None):
    """xr / xi may be None when xp (input_grid) is given; gp: a grad_grid shared with conv_dgrad.
    bias_out: a list the BIAS gradient (per-channel sums of gr [, gi], float32) is appended to when the kernel that runs
    produces it on the way (the generic kernel: one more GEMM column); left empty otherwise -- the caller then sums."""
    lib = _lib.load()
    cplx = gi is not None
    if geom[0] == 0:                       # empty batch: the gradient is zero
        dwr = torch.zeros(w_shape, dtype=torch.float32, device=gr.device)
        return dwr, (torch.zeros_like(dwr) if cplx else None)
    if gr.dtype in (torch.bfloat16, torch.float32) and _rows_wgrad_ok(geom, cplx, gr.dtype):
        xp = input_grid(xr, xi, geom) if xp is None else xp
        gp = grad_grid(gr, gi, geom) if gp is None else gp
        out = _wgrad_rows(gp, xp, geom, w_shape, emul)
        if out is not None:
            return out
    if xr is None:
        raise CplxAmdError("conv_wgrad: the channels-last kernel refused a shape it was planned for")
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
    db = None
    if bias_out is not None:
        db = torch.empty((2 if cplx else 1), geom[2], dtype=torch.float32, device=gr.device)
        bias_out.extend(db[i] for i in range(db.shape[0]))
    call("cplxamd_conv2d_wgrad_bias", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(emul), ptr(dwr), ptr(dwi), ptr(db),
         ptr(db[1]) if (db is not None and cplx) else None, geom, dtype_code(gr), ptr(ws), ws.numel(), stream_ptr())
    return dwr, dwi


def _shared_grad_grid(gr, gi, geom, need_dx, need_dw):
    """One gradient grid for both backward kernels (None if neither takes the rows path)."""
    if gr.dtype not in (torch.bfloat16, torch.float32):
        return None
    if (need_dx and _rows_dgrad_ok(geom, gr.dtype)) or (need_dw and _rows_wgrad_ok(geom, gi is not None, gr.dtype)):
        return grad_grid(gr, gi, geom)
    return None


def chansum(g):
    """sum over (batch, spatial) per channel of an NCHW tensor -> float32 [C]."""
    B, C = g.shape[0], g.shape[1]
    if g.numel() == 0:
        return torch.zeros(C, dtype=torch.float32, device=g.device)
    S = g.numel() // (B * C)
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = _scratch(g.device, 64 * C * 8)
    call("cplxamd_chansum", ptr(g), ptr(out), B, C, S, dtype_code(g), ptr(ws), stream_ptr())
    return out


def chansum2(gr, gi):
    """The complex bias gradient: per-channel sums of both NCHW planes, both planes per launch -> float32 ([C], [C])."""
    B, C = gr.shape[0], gr.shape[1]
    if gr.numel() == 0:
        return (torch.zeros(C, dtype=torch.float32, device=gr.device),) * 2
    S = gr.numel() // (B * C)
    out = torch.empty(2, C, dtype=torch.float32, device=gr.device)
    ws = _scratch(gr.device, 2 * 64 * C * 8)
    call("cplxamd_chansum2", ptr(gr), ptr(gi), ptr(out[0]), ptr(out[1]), B, C, S, dtype_code(gr), ptr(ws), stream_ptr())
    return out[0], out[1]


def _cl_backward(ctx, gr, gi, xr, xi, wcr, wci):
    """Backward of the channels-last forward: xr / xi are the saved channels-last inputs."""
    need = ctx.needs_input_grad
    geom = ctx.geom
    dxr = dxi = dwr = dwi = dbr = dbi = None
    hint_r, hint_i = ops.colsum_hint(gr), ops.colsum_hint(gi)
    dw_hint = ops.wgrad_hint(gr, gi, ctx)      # (bn.py: the batch-norm backward formed gr / gi inside this layer's wgrad kernel)
    gr, gi = to_channels_last(gr), to_channels_last(gi)
    if need[0] or need[1]:
        if _cl_ok(geom, dgrad=True):
            dxr, dxi = cl_conv(gr, gi, wcr, wci, None, None, geom, dgrad=True)
            if ctx.x_planar:             # the input was a plain contiguous tensor: hand its gradient back in that layout
                dxr, dxi = from_channels_last(dxr), from_channels_last(dxi)   # (autograd would do it with a slow copy)
        else:
            dxr, dxi = conv_dgrad(from_channels_last(gr), from_channels_last(gi), wcr, wci, geom, ctx.xshape)
    if need[2] or need[3]:
        if dw_hint is not None:
            dwr, dwi = dw_hint
        elif _cl_wgrad_ok(geom):
            dwr, dwi = cl_wgrad(gr, gi, xr, xi, geom, ctx.wshape)
        else:
            dwr, dwi = conv_wgrad(from_channels_last(gr), from_channels_last(gi), from_channels_last(xr),
                                  from_channels_last(xi), geom, ctx.wshape)
    if ctx.has_bias and (need[4] or need[5]):
        B, Co, H, W = gr.shape
        dbr = hint_r if hint_r is not None else ops.colsum(gr.permute(0, 2, 3, 1).reshape(B * H * W, Co))
        dbi = hint_i if hint_i is not None else ops.colsum(gi.permute(0, 2, 3, 1).reshape(B * H * W, Co))
    return dxr, dxi, dwr, dwi, dbr, dbi, None, None, None, None, None


# ---- float32 layers on IEEE-half pieces (x3.py 'x2'; csrc/conv_cl2_f16.hip, conv_cl_wgrad_f16.hip) ---------------------- #
# A float32 3 x 3 convolution (stride 1, groups 1, dilation 1, padding <= same, channel counts multiples of 64) runs as
# three piece products on the half matrix pipe: the channels-last planes, read as [B H W][C] matrices, are cut into
# [h1|h0] rows (x3.split_planes: one power-of-two scale per operand), forward and data gradient are two launches of the
# 2-d-patch kernel -- h0 * w1 on the h0 window of the rows, then [h1|h0] * [w0|w0] accumulated into the float32 output --
# and the weight gradient one launch on ([g1|g0], [x1|x0]) whose four piece blocks give (g0 x1) + (g1 x0) + (g0 x0).
_X2_CONV_MIN_FLOP = float(1 << 33)
_X2_BYTES_MAX = 0xF0000000


def _x2_conv_kind(geom, *tensors):
    from . import x3
    mode = x3.get_fp32_mode()
    if mode not in ("auto", "x2") or (mode == "auto" and x3.AUTO_KIND != "x2"):
        return None
    if any(t is None or t.dtype != torch.float32 or not t.is_cuda for t in tensors):
        return None
    B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups = (geom[i] for i in range(14))
    if not _CL_ENABLED or sh != 1 or sw != 1 or groups != 1 or KH != 3 or KW != 3 or dh != 1 or dw != 1 or ph > 1 or pw > 1:
        return None
    if Ci % 64 or Co % 64 or B <= 0 or H + 2 * ph - 2 <= 0 or W + 2 * pw - 2 <= 0:
        return None
    if H * W * 2 * max(Ci, Co) * 2 >= _X2_BYTES_MAX:          # one image of pieces must fit a buffer descriptor
        return None
    if mode == "auto" and 8.0 * B * H * W * Ci * Co * 9 < _X2_CONV_MIN_FLOP:
        return None
    return "x2"


def _rows(t):
    """channels-last [B, C, H, W] -> its storage as a [B H W, C] matrix (a view)."""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


_X2_ONE_LAUNCH = os.environ.get("CPLXAMD_X2_ONE", "1") != "0"    # (A/B: 0 = the two launches of the first x2 form)


def _x2_weight_packs(wr, wi, dgrad):
    """Packed LDS images of the weights, and the weights' scale.  One launch (the contraction window wraps around the
    [h1|h0] pixel as [h0|h1|h0]): (None, pack of [w1|w0|w0] -- 3 C channels --, scale); two launches: (pack of w1 -- C
    channels --, pack of [w0|w0] -- 2 C channels --, scale)."""
    from . import x3
    Co, Ci, KH, KW = wr.shape
    pr, pi = x3.split_planes((wr.reshape(Co, -1), wi.reshape(Co, -1)), kind="x2")          # [Co, 2 Ci 9] = [h1|h0]
    n = Ci * KH * KW
    cut = lambda p: (p.t[:, :n].reshape(Co, Ci, KH, KW), p.t[:, n:].reshape(Co, Ci, KH, KW))  # noqa: E731  (w1, w0)
    (w1r, w0r), (w1i, w0i) = cut(pr), cut(pi)
    cat = 0 if dgrad else 1                            # the contraction channels: Co for the data gradient, Ci forward
    if _X2_ONE_LAUNCH:
        return None, _cl_pack(torch.cat([w1r, w0r, w0r], cat).contiguous(), torch.cat([w1i, w0i, w0i], cat).contiguous(),
                              dgrad), pr.scale
    small = _cl_pack(w1r.contiguous(), w1i.contiguous(), dgrad)
    big = _cl_pack(torch.cat([w0r, w0r], cat).contiguous(), torch.cat([w0i, w0i], cat).contiguous(), dgrad)
    return small, big, pr.scale


def _x2_conv(pieces, wr, wi, br, bi, geom, dgrad):
    """Forward (or data gradient) from the [h1|h0] pieces of the input planes ([B H W, 2 C] half matrices)."""
    B, Ci, Co, H, W, KH, KW = (geom[i] for i in range(7))
    C, N = (Co, Ci) if dgrad else (Ci, Co)
    ph, pw = geom[9], geom[10]
    Ho, Wo = H + 2 * ph - 2, W + 2 * pw - 2
    Hin, Win = (Ho, Wo) if dgrad else (H, W)
    oshape = (B, N, H, W) if dgrad else (B, N, Ho, Wo)
    pr, pi = pieces
    dev = pr.t.device
    yr = torch.empty(oshape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    yi = torch.empty_like(yr)
    small, big, wscale = _x2_weight_packs(wr, wi, dgrad)
    ws = _scratch(dev, int(_lib.load().cplxamd_conv2d_cl_ws_bytes(N)))
    flags = launch_flags()
    per = Hin * Win * 2 * C * 2                                   # bytes of one image of pieces
    step = max(1, min(B, (_X2_BYTES_MAX - 1) // per))
    step = -(-B // (-(-B // step)))                      # equal chunks
    opix = oshape[2] * oshape[3] * N
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        xr_ = pr.t[b0 * Hin * Win:]
        xi_ = pi.t[b0 * Hin * Win:]
        o_r = yr.permute(0, 2, 3, 1).reshape(-1)[b0 * opix:]
        o_i = yi.permute(0, 2, 3, 1).reshape(-1)[b0 * opix:]
        if small is None:
            # one launch: the window [h0|h1|h0] (starts C channels into the [h1|h0] pixel, wraps) * [w1|w0|w0] (+ bias)
            call("cplxamd_conv2d_cl2h_wrap_fl", ptr(xr_), ptr(xi_), 2 * C, C, ptr(big), ptr(br), ptr(bi), ptr(o_r), ptr(o_i), 0,
                 ptr(pr.scale), ptr(wscale), nb, H, W, 3 * C, N, ph, pw, int(dgrad), ptr(ws), ws.numel(), flags, stream_ptr())
            continue
        # launch 1: h0 (the second half of every row) * w1 (+ bias); launch 2: [h1|h0] * [w0|w0], accumulated
        h0r, h0i = xr_.reshape(-1)[C:], xi_.reshape(-1)[C:]
        call("cplxamd_conv2d_cl2h_fl", ptr(h0r), ptr(h0i), 2 * C, ptr(small), ptr(br), ptr(bi), ptr(o_r), ptr(o_i), 0,
             ptr(pr.scale), ptr(wscale), nb, H, W, C, N, ph, pw, int(dgrad), ptr(ws), ws.numel(), flags, stream_ptr())
        call("cplxamd_conv2d_cl2h_fl", ptr(xr_), ptr(xi_), 2 * C, ptr(big), None, None, ptr(o_r), ptr(o_i), 1,
             ptr(pr.scale), ptr(wscale), nb, H, W, 2 * C, N, ph, pw, int(dgrad), ptr(ws), ws.numel(), flags, stream_ptr())
    return yr, yi


def _x2_conv_wgrad(gp, xp, geom, w_shape):
    """dW from the pieces of the output gradient and of the input: one launch per batch chunk on (2 Co, 2 Ci) channels."""
    B, Ci, Co, H, W, KH, KW = (geom[i] for i in range(7))
    ph, pw = geom[9], geom[10]
    Ho, Wo = H + 2 * ph - 2, W + 2 * pw - 2
    dev = gp[0].t.device
    lim = (1 << 32) - 64
    step = max(1, min(B, (lim // (2 * max(Ci, Co) * 2) - ph * W - 64) // (H * W)))
    step = -(-B // (-(-B // step)))                      # equal chunks
    flags = launch_flags()
    tot_r = tot_i = None
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        ws = _scratch(dev, int(_lib.load().cplxamd_conv2d_clh_wgrad_ws_bytes(nb, H, W, 2 * Ci, 2 * Co)))
        dwr = torch.empty(2 * Co, 2 * Ci, KH, KW, dtype=torch.float32, device=dev)
        dwi = torch.empty_like(dwr)
        g_r, g_i = gp[0].t[b0 * Ho * Wo:], gp[1].t[b0 * Ho * Wo:]
        x_r, x_i = xp[0].t[b0 * H * W:], xp[1].t[b0 * H * W:]
        # (the block g1 x1 -- rows < Co, columns < Ci -- is none of the three piece products: not computed, not read below)
        call("cplxamd_conv2d_clh_wgrad_skip_fl", ptr(g_r), ptr(g_i), ptr(x_r), ptr(x_i), ptr(dwr), ptr(dwi), nb, H, W, 2 * Ci,
             2 * Co, KH, KW, 1, 1, ph, pw, Co, Ci, ptr(ws), ws.numel(), flags, stream_ptr())
        dwr[:Co, :Ci] = 0
        dwi[:Co, :Ci] = 0
        tot_r, tot_i = (dwr, dwi) if tot_r is None else (tot_r + dwr, tot_i + dwi)
    alpha = gp[0].scale[1] * xp[0].scale[1]
    pick = lambda t: ((t[Co:, :Ci] + t[:Co, Ci:]) + t[Co:, Ci:]) * alpha  # noqa: E731  (g0 x1) + (g1 x0) + (g0 x0)
    return pick(tot_r).reshape(w_shape).contiguous(), pick(tot_i).reshape(w_shape).contiguous()


class CplxConv2dFn(torch.autograd.Function):
    """Zero-padded complex conv (A.1 algebra with cross-correlation)."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, stride, padding, dilation, groups, moments=False):
        require_device(xr, xi, wr, wi, br, bi)
        geom, oshape = _geom(xr.shape, wr.shape, stride, padding, dilation, groups)
        b = (None, None) if br is None else (br.contiguous(), bi.contiguous())
        ctx.x2 = _x2_conv_kind(geom, xr, xi, wr, wi) is not None
        if ctx.x2:                       # float32 on IEEE-half pieces, channels-last in and out (see _x2_conv)
            from . import x3
            ctx.cl = False
            ctx.x_planar = xr.is_contiguous() and not xr.is_contiguous(memory_format=torch.channels_last)
            xr, xi = (t.contiguous(memory_format=torch.channels_last) for t in (xr, xi))
            xp = x3.split_planes((_rows(xr), _rows(xi)), kind="x2")
            wr_, wi_ = wr.contiguous(), wi.contiguous()
            yr, yi = _x2_conv(xp, wr_, wi_, b[0], b[1], geom, dgrad=False)
            ctx.save_for_backward(xp[0].t, xp[1].t, xp[0].scale, wr_, wi_)   # (the pieces: the float32 input is not needed again)
            ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, br is not None, wr.shape, xr.shape
            return yr, yi
        # (a layer whose weight gradient cannot run channels-last -- image width not a multiple of 32 -- stays on the
        #  planar path as a whole: converting both operands back for it costs more than the forward saves)
        ctx.cl = xr.dtype == torch.bfloat16 and xi.dtype == torch.bfloat16 and _cl_ok(geom) and (
            _cl_wgrad_ok(geom) or not (wr.requires_grad or wi.requires_grad))
        if ctx.cl:                       # channels-last in, channels-last out: no layout copies between such layers
            ctx.x_planar = xr.is_contiguous() and not xr.is_contiguous(memory_format=torch.channels_last)
            xr, xi = to_channels_last(xr), to_channels_last(xi)
            wcr, wci = ops.cast(wr.contiguous(), xr.dtype), ops.cast(wi.contiguous(), xr.dtype)
            yr, yi = cl_conv(xr, xi, wcr, wci, b[0], b[1], geom, moments=moments)
            ctx.save_for_backward(xr, xi, wcr, wci)
            ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, br is not None, wr.shape, xr.shape
            return yr, yi
        xr, xi = xr.contiguous(), xi.contiguous()
        wcr, wci = ops.cast(wr.contiguous(), xr.dtype), ops.cast(wi.contiguous(), xr.dtype)
        yr, yi, xp = conv_fwd(xr, xi, wcr, wci, b[0], b[1], geom, oshape, keep_grid=True)
        # keep the channels-last input for the weight gradient instead of the planar one
        ctx.grid = xp is not None and _rows_wgrad_ok(geom, True, xr.dtype)
        ctx.save_for_backward(*(xp if ctx.grid else (xr, xi)), wcr, wci)
        ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, br is not None, wr.shape, xr.shape
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        need = ctx.needs_input_grad
        dxr = dxi = dwr = dwi = dbr = dbi = None
        if ctx.x2:
            from . import x3
            xtr, xti, xsc, wr_, wi_ = ctx.saved_tensors
            hint_r, hint_i = ops.colsum_hint(gr), ops.colsum_hint(gi)      # (a batch-norm backward's column sums of gr / gi)
            gscale = ops.scale_hint(gr, gi)                                # (... and the scale of their largest magnitude)
            gr, gi = (t.contiguous(memory_format=torch.channels_last) for t in (gr, gi))
            gp = x3.split_planes((_rows(gr), _rows(gi)), kind="x2", scale=gscale)
            if need[0] or need[1]:
                dxr, dxi = _x2_conv(gp, wr_, wi_, None, None, ctx.geom, dgrad=True)
                if ctx.x_planar:
                    dxr, dxi = dxr.contiguous(), dxi.contiguous()
            if need[2] or need[3]:
                xp = (x3.Pieces(xtr, "x2", xsc, 2), x3.Pieces(xti, "x2", xsc, 2))
                dwr, dwi = _x2_conv_wgrad(gp, xp, ctx.geom, ctx.wshape)
            if ctx.has_bias and (need[4] or need[5]):
                dbr = hint_r if hint_r is not None else ops.colsum(_rows(gr))
                dbi = hint_i if hint_i is not None else ops.colsum(_rows(gi))
            return dxr, dxi, dwr, dwi, dbr, dbi, None, None, None, None, None
        xr, xi, wcr, wci = ctx.saved_tensors
        if ctx.cl:
            return _cl_backward(ctx, gr, gi, xr, xi, wcr, wci)
        gr, gi = gr.contiguous(), gi.contiguous()
        gp = _shared_grad_grid(gr, gi, ctx.geom, need[0] or need[1], need[2] or need[3])
        if need[0] or need[1]:
            dxr, dxi = conv_dgrad(gr, gi, wcr, wci, ctx.geom, ctx.xshape, gp=gp)
        want_b = ctx.has_bias and (need[4] or need[5])
        bsum = [] if want_b else None                 # filled by the weight-gradient kernel when it can carry the sums
        if need[2] or need[3]:
            if ctx.grid:
                dwr, dwi = conv_wgrad(gr, gi, None, None, ctx.geom, ctx.wshape, xp=(xr, xi), gp=gp, bias_out=bsum)
            else:
                dwr, dwi = conv_wgrad(gr, gi, xr, xi, ctx.geom, ctx.wshape, gp=gp, bias_out=bsum)
        if want_b:
            dbr, dbi = bsum if bsum else chansum2(gr, gi)
        return dxr, dxi, dwr, dwi, dbr, dbi, None, None, None, None, None


class RealConv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups):
        require_device(x, w, b)
        geom, oshape = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        ctx.cl = _cl_layer_ok(geom, x)
        if ctx.cl:
            ctx.x_planar = x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)
            x = to_channels_last(x)
            wc = ops.cast(w.contiguous(), x.dtype)
            y = cl_conv_real(x, wc, None if b is None else b.contiguous(), geom)
            ctx.save_for_backward(x, wc)
            ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, b is not None, w.shape, x.shape
            return y
        x = x.contiguous()
        wc = ops.cast(w.contiguous(), x.dtype)
        y, _, xp = conv_fwd(x, None, wc, None, None if b is None else b.contiguous(), None, geom,
                            oshape, keep_grid=True)
        ctx.grid = xp is not None and _rows_wgrad_ok(geom, False, x.dtype)
        ctx.save_for_backward(xp[0] if ctx.grid else x, wc)
        ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, b is not None, w.shape, x.shape
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, wc = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx = dw = db = None
        if ctx.cl:
            hint = ops.colsum_hint(g)
            g = to_channels_last(g)
            if need[0]:
                dx = cl_conv_real(g, wc, None, ctx.geom, dgrad=True)
                if ctx.x_planar:
                    dx = from_channels_last(dx)
            if need[1]:
                dw = cl_wgrad_real(g, x, ctx.geom, ctx.wshape)
            if ctx.has_bias and need[2]:
                B, Co, H, W = g.shape
                db = hint if hint is not None else ops.colsum(g.permute(0, 2, 3, 1).reshape(B * H * W, Co))
            return dx, dw, db, None, None, None, None
        g = g.contiguous()
        gp = _shared_grad_grid(g, None, ctx.geom, need[0], need[1])
        if need[0]:
            dx, _ = conv_dgrad(g, None, wc, None, ctx.geom, ctx.xshape, gp=gp)
        want_b = ctx.has_bias and need[2]
        bsum = [] if want_b else None
        if need[1]:
            if ctx.grid:
                dw, _ = conv_wgrad(g, None, None, None, ctx.geom, ctx.wshape, xp=(x, None), gp=gp, bias_out=bsum)
            else:
                dw, _ = conv_wgrad(g, None, x, None, ctx.geom, ctx.wshape, gp=gp, bias_out=bsum)
        if want_b:
            db = bsum[0] if bsum else chansum(g)
        return dx, dw, db, None, None, None, None


def _circular_pad(t, padding):
    """symmetric_circular_padding (cplxmodule/cplx.py:701-714): ((p+1)//2, p//2) per spatial
    dim; F.pad's tuple starts at the LAST dim, the reference feeds `padding` in that order."""
    pads = []
    for p in _pair(padding):
        pads.extend(((p + 1) // 2, p // 2))
    return F.pad(t, tuple(pads), mode="circular")


def cplx_conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                padding_mode="zeros", training=True):
    xr, xi = input.real, input.imag
    if padding_mode == "circular":
        xr, xi, padding = _circular_pad(xr, padding), _circular_pad(xi, padding), 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    if _pointwise_cl(xr, xi, weight.real, stride, padding, groups):
        # a 1 x 1 convolution on channels-last planes IS the linear layer on [B H W, C] rows: the complex GEMM kernels
        # (forward, both gradients, bias column sums), no copies, channels-last out
        B, C, H, W = xr.shape
        Co = weight.real.shape[0]
        rows = lambda t: ToChannelsLastFn.apply(t).permute(0, 2, 3, 1)  # noqa: E731  ([B, H, W, C] view of the storage)
        yr, yi = ops.CplxLinearFn.apply(rows(xr), rows(xi), weight.real.reshape(Co, C), weight.imag.reshape(Co, C), br, bi)
        return Cplx(yr.permute(0, 3, 1, 2), yi.permute(0, 3, 1, 2))
    wkey = weight.real
    yr, yi = CplxConv2dFn.apply(xr, xi, wkey, weight.imag, br, bi, stride, padding,
                                dilation, groups, _MOMENTS and training and moments_wanted(wkey, spend=True))
    # (training=False -- an evaluation-mode CplxConv2d: no batch-norm layer will read batch statistics, so an armed
    #  request, permanent ones included, neither runs the +10 % epilogue nor spends a credit: ADVICE r05)
    if _MOMENTS and yr.dtype == torch.bfloat16:
        yr._cplxamd_conv_src = weakref.ref(wkey)       # (a batch-norm layer that consumes yr arms the moments epilogue)
    return Cplx(yr, yi)


def _pointwise_cl(xr, xi, w, stride, padding, groups):
    """1 x 1, stride 1, no padding, one group, bf16 images that are channels-last already or large enough to pay for
    one conversion."""
    if not _CL_ENABLED or xr.dim() != 4 or xr.dtype != torch.bfloat16 or xi.dtype != torch.bfloat16 or not xr.is_cuda:
        return False
    if tuple(w.shape[2:]) != (1, 1) or _pair(stride) != (1, 1) or _pair(padding) != (0, 0) or groups != 1:
        return False
    B, C, H, W = xr.shape
    if B * H * W == 0 or C % 8 or w.shape[0] % 8:
        return False
    is_cl = xr.is_contiguous(memory_format=torch.channels_last) and not xr.is_contiguous()
    return is_cl or _CL_FORCE or 8.0 * B * H * W * C * w.shape[0] >= _CL_MIN_FLOP


class CplxConv2dLRTFn(torch.autograd.Function):
    """mu conv + variance conv + noise injection; backward per SURVEY A.2 with conv."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i, seed, offset, stride, padding,
                dilation, groups):
        require_device(xr, xi, wr, wi, br, bi, ls2, eps_r, eps_i)
        geom, oshape = _geom(xr.shape, wr.shape, stride, padding, dilation, groups)
        ctx.cl = _cl_layer_ok(geom, xr, xi)
        if ctx.cl:      # channels-last end to end: mean conv, variance conv, noise injection all on [B][H][W][C] planes
            ctx.x_planar = xr.is_contiguous() and not xr.is_contiguous(memory_format=torch.channels_last)
            xr, xi = to_channels_last(xr), to_channels_last(xi)
        else:
            xr, xi = xr.contiguous(), xi.contiguous()
        dt = xr.dtype
        wcr, wci = ops.cast(wr.contiguous(), dt), ops.cast(wi.contiguous(), dt)
        b = (None, None) if br is None else (br.contiguous(), bi.contiguous())
        a = ops.abs2(xr, xi)
        S = ops.exp(ls2.contiguous(), out_dtype=dt)
        if ctx.cl:
            mur, mui = cl_conv(xr, xi, wcr, wci, b[0], b[1], geom)
            s2 = cl_conv_real(a, S, None, geom)
        else:
            mur, mui = conv_fwd(xr, xi, wcr, wci, b[0], b[1], geom, oshape)
            s2, _ = conv_fwd(a, None, S, None, None, None, geom, oshape)
        eps = None if eps_r is None else (eps_r, eps_i)
        yr, yi = ops.reparam_fwd(mur, mui, s2, eps, seed, offset, inplace=True)
        ctx.save_for_backward(xr, xi, wcr, wci, ls2, s2, a, S, eps_r, eps_i)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, br is not None, wr.shape
        ctx.seed, ctx.offset = seed, offset
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        xr, xi, wcr, wci, ls2, s2, a, S, eps_r, eps_i = ctx.saved_tensors
        need = ctx.needs_input_grad
        eps = None if eps_r is None else (eps_r, eps_i)
        dxr = dxi = dwr = dwi = dbr = dbi = dls2 = None
        if ctx.cl:
            geom = ctx.geom
            hint_r, hint_i = ops.colsum_hint(gr), ops.colsum_hint(gi)
            gr, gi = to_channels_last(gr), to_channels_last(gi)
            want_b = ctx.has_bias and (need[4] or need[5])
            if want_b and (hint_r is None or hint_i is None):
                # the bias gradient (per-channel sums of G) comes out of the pass that reads G for d s2
                Bn, Co_, Hn, Wn = gr.shape
                gs2, hint_r, hint_i = ops.reparam_bwd(gr, gi, s2, eps, ctx.seed, ctx.offset, out_dtype=xr.dtype,
                                                      bias_sums=(Bn * Hn * Wn, Co_))
            else:
                gs2 = ops.reparam_bwd(gr, gi, s2, eps, ctx.seed, ctx.offset, out_dtype=xr.dtype)
            if need[0] or need[1]:
                ga = cl_conv_real(gs2, S, None, geom, dgrad=True)
                dxr, dxi = cl_conv_lrt_dx(gr, gi, wcr, wci, geom, xr, xi, ga)
                if ctx.x_planar:
                    dxr, dxi = from_channels_last(dxr), from_channels_last(dxi)
            if need[2] or need[3]:
                dwr, dwi = cl_wgrad(gr, gi, xr, xi, geom, ctx.wshape)
            if want_b:
                dbr, dbi = hint_r, hint_i
            if need[6]:
                dls2 = cl_wgrad_real(gs2, a, geom, ctx.wshape, emul=ls2.contiguous(), emul_exp=True)
            return (dxr, dxi, dwr, dwi, dbr, dbi, dls2) + (None,) * 8
        gr, gi = gr.contiguous(), gi.contiguous()
        gs2 = ops.reparam_bwd(gr, gi, s2, eps, ctx.seed, ctx.offset, out_dtype=xr.dtype)
        if need[0] or need[1]:
            dxr, dxi = conv_dgrad(gr, gi, wcr, wci, ctx.geom, xr.shape)
            ga, _ = conv_dgrad(gs2, None, S, None, ctx.geom, xr.shape)
            ops.lrt_dx_accum(dxr, dxi, xr, xi, ga)
        if need[2] or need[3]:
            dwr, dwi = conv_wgrad(gr, gi, xr, xi, ctx.geom, ctx.wshape)
        if ctx.has_bias and (need[4] or need[5]):
            dbr, dbi = chansum2(gr, gi)
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
        geom, oshape = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        ctx.cl = _cl_layer_ok(geom, x)
        if ctx.cl:
            ctx.x_planar = x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)
            x = to_channels_last(x)
        else:
            x = x.contiguous()
        dt = x.dtype
        wc = ops.cast(w.contiguous(), dt)
        a = ops.abs2(x)
        S = ops.exp(ls2.contiguous(), out_dtype=dt)
        if ctx.cl:
            mu = cl_conv_real(x, wc, None if b is None else b.contiguous(), geom)
            s2 = cl_conv_real(a, S, None, geom)
        else:
            mu, _ = conv_fwd(x, None, wc, None, None if b is None else b.contiguous(), None, geom, oshape)
            s2, _ = conv_fwd(a, None, S, None, None, None, geom, oshape)
        y, _ = ops.reparam_fwd(mu, None, s2, eps, seed, offset, inplace=True)
        ctx.save_for_backward(x, wc, ls2, s2, a, S, eps)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, b is not None, w.shape
        ctx.seed, ctx.offset = seed, offset
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, wc, ls2, s2, a, S, eps = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx = dw = db = dls2 = None
        if ctx.cl:
            geom = ctx.geom
            hint = ops.colsum_hint(g)
            g = to_channels_last(g)
            want_b = ctx.has_bias and need[2]
            if want_b and hint is None:
                Bn, Co_, Hn, Wn = g.shape
                gs2, hint, _ = ops.reparam_bwd(g, None, s2, eps, ctx.seed, ctx.offset, out_dtype=x.dtype,
                                               bias_sums=(Bn * Hn * Wn, Co_))
            else:
                gs2 = ops.reparam_bwd(g, None, s2, eps, ctx.seed, ctx.offset, out_dtype=x.dtype)
            if need[0]:
                dx = cl_conv_real(g, wc, None, geom, dgrad=True)
                ga = cl_conv_real(gs2, S, None, geom, dgrad=True)
                ops.lrt_dx_accum(dx, None, x, None, ga)
                if ctx.x_planar:
                    dx = from_channels_last(dx)
            if need[1]:
                dw = cl_wgrad_real(g, x, geom, ctx.wshape)
            if want_b:
                db = hint
            if need[3]:
                dls2 = cl_wgrad_real(gs2, a, geom, ctx.wshape, emul=ls2.contiguous(), emul_exp=True)
            return (dx, dw, db, dls2) + (None,) * 7
        g = g.contiguous()
        gs2 = ops.reparam_bwd(g, None, s2, eps, ctx.seed, ctx.offset, out_dtype=x.dtype)
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


# ------------------------------------------------------------------------------------------ #
#  1-d convolution: the 2-d kernels on a height-1 image (cplx.conv1d, cplxmodule/cplx.py:803-819) #
# ------------------------------------------------------------------------------------------ #
def _one(v):
    return v if isinstance(v, int) else v[0]


class _Lift1d:
    """A 1-d conv layer seen as the equivalent 2-d one: weights / log_sigma2 get a unit height
    (views: gradients flow back to the 3-d parameters), the 1-d hyper-parameters become (1, s),
    (0, p), (1, d).  Everything else is the wrapped layer's."""

    def __init__(self, layer):
        self._layer = layer

    def __getattr__(self, name):
        return getattr(self._layer, name)

    @property
    def weight(self):
        w = self._layer.weight
        return Cplx(w.real.unsqueeze(2), w.imag.unsqueeze(2)) if isinstance(w, Cplx) else w.unsqueeze(2)

    @property
    def log_sigma2(self):
        return self._layer.log_sigma2.unsqueeze(2)

    stride = property(lambda self: (1, _one(self._layer.stride)))
    padding = property(lambda self: (0, _one(self._layer.padding)))
    dilation = property(lambda self: (1, _one(self._layer.dilation)))


def _up(x):
    return Cplx(x.real.unsqueeze(2), x.imag.unsqueeze(2)) if isinstance(x, Cplx) else x.unsqueeze(2)


def _down(y):
    return Cplx(y.real.squeeze(2), y.imag.squeeze(2)) if isinstance(y, Cplx) else y.squeeze(2)


def cplx_conv1d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                padding_mode="zeros"):
    """Complex 1-d cross-correlation on [B, C, L]."""
    if padding_mode == "circular":
        p = _one(padding)
        pads = ((p + 1) // 2, p // 2)
        input = Cplx(F.pad(input.real, pads, mode="circular"), F.pad(input.imag, pads, mode="circular"))
        padding = 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    w2 = Cplx(weight.real.unsqueeze(2), weight.imag.unsqueeze(2))
    y = cplx_conv2d(_up(input), w2, bias, (1, _one(stride)), (0, _one(padding)), (1, _one(dilation)), groups)
    return _down(y)


def cplx_conv1d_lrt(layer, input, eps=None):
    return _down(cplx_conv2d_lrt(_Lift1d(layer), _up(input), None if eps is None else _up(eps)))


def real_conv1d_layer(layer, input, eps=None):
    return _down(real_conv2d_layer(_Lift1d(layer), _up(input), None if eps is None else _up(eps)))


# ------------------------------------------------------------------------------------------ #
#  transposed convolution = the data-gradient kernels run forward                              #
# ------------------------------------------------------------------------------------------ #
def _transpose_out_shape(x_shape, w_shape, stride, padding, output_padding, dilation, groups):
    B, Cin, H, W = x_shape
    _, Cog, KH, KW = w_shape
    (sh, sw), (ph, pw), (oh, ow), (dh, dw) = (_pair(v) for v in (stride, padding, output_padding, dilation))
    return (B, Cog * groups, (H - 1) * sh - 2 * ph + dh * (KH - 1) + oh + 1,
            (W - 1) * sw - 2 * pw + dw * (KW - 1) + ow + 1)


class CplxConvTranspose2dFn(torch.autograd.Function):
    """y = x (*)^T W + b without conjugation (cplx.conv_transposend_naive, cplxmodule/cplx.py:860-873).
    The transposed correlation IS the adjoint of `Conv_V`, which the dgrad kernels compute as
    g -> g (*)^T conj(V); with V = conj(W) that is the forward pass here.  Backward, by the same
    adjointness:  dx = Conv_V(gy) (the forward conv kernels),  dV = wgrad(g_out = x, input = gy),
    dW = conj(dV),  db = sum gy."""

    @staticmethod
    def forward(ctx, xr, xi, wr, wi, br, bi, stride, padding, output_padding, dilation, groups):
        require_device(xr, xi, wr, wi, br, bi)
        xr, xi = xr.contiguous(), xi.contiguous()
        if wr.shape[0] != xr.shape[1]:
            raise ValueError(f"expected {wr.shape[0]} input channels, got {xr.shape[1]}")
        yshape = _transpose_out_shape(xr.shape, wr.shape, stride, padding, output_padding, dilation, groups)
        geom, oshape = _geom(yshape, wr.shape, stride, padding, dilation, groups)
        if tuple(oshape) != tuple(xr.shape):
            raise ValueError("output_padding must be smaller than either stride or dilation")
        vr, vi = ops.cast(wr.contiguous(), xr.dtype), ops.cast((-wi).contiguous(), xr.dtype)
        yr, yi = conv_dgrad(xr, xi, vr, vi, geom, yshape)
        if br is not None:
            yr, yi = yr + br.view(1, -1, 1, 1).to(yr.dtype), yi + bi.view(1, -1, 1, 1).to(yi.dtype)
        ctx.save_for_backward(xr, xi, vr, vi)
        ctx.geom, ctx.has_bias, ctx.wshape, ctx.xshape = geom, br is not None, wr.shape, xr.shape
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        xr, xi, vr, vi = ctx.saved_tensors
        gr, gi = gr.contiguous(), gi.contiguous()
        need = ctx.needs_input_grad
        dxr = dxi = dwr = dwi = dbr = dbi = None
        if need[0] or need[1]:
            dxr, dxi = conv_fwd(gr, gi, vr, vi, None, None, ctx.geom, ctx.xshape)
        if need[2] or need[3]:
            dwr, dvi = conv_wgrad(xr, xi, gr, gi, ctx.geom, ctx.wshape)
            dwi = -dvi
        if ctx.has_bias and (need[4] or need[5]):
            dbr, dbi = chansum2(gr, gi)
        return (dxr, dxi, dwr, dwi, dbr, dbi) + (None,) * 5


def cplx_conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1,
                          dilation=1, padding_mode="zeros"):
    xr, xi = input.real, input.imag
    if padding_mode == "circular":     # the reference pads the INPUT circularly, then runs padding 0
        xr, xi, padding = _circular_pad(xr, padding), _circular_pad(xi, padding), 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr, yi = CplxConvTranspose2dFn.apply(xr, xi, weight.real, weight.imag, br, bi, stride, padding,
                                         output_padding, dilation, groups)
    return Cplx(yr, yi)


def cplx_conv_transpose1d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1,
                          dilation=1, padding_mode="zeros"):
    if padding_mode == "circular":
        p = _one(padding)
        pads = ((p + 1) // 2, p // 2)
        input = Cplx(F.pad(input.real, pads, mode="circular"), F.pad(input.imag, pads, mode="circular"))
        padding = 0
    elif padding_mode != "zeros":
        raise ValueError("padding_mode must be 'zeros' or 'circular'.")
    w2 = Cplx(weight.real.unsqueeze(2), weight.imag.unsqueeze(2))
    y = cplx_conv_transpose2d(_up(input), w2, bias, (1, _one(stride)), (0, _one(padding)),
                              (0, _one(output_padding)), groups, (1, _one(dilation)))
    return _down(y)


# float64: a parity mode on its own kernels (f64.py; ops.Route dispatches on the dtype of the first tensor argument)
CplxConv2dFn = ops.Route(CplxConv2dFn, "cplx_conv2d")
RealConv2dFn = ops.Route(RealConv2dFn, "real_conv2d")
CplxConv2dLRTFn = ops.Route(CplxConv2dLRTFn, "cplx_conv2d_lrt")
RealConv2dLRTFn = ops.Route(RealConv2dLRTFn, "real_conv2d_lrt")
