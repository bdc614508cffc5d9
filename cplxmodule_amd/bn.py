"""Complex batch-norm autograd Function over the 2-pass moment / apply kernels (csrc/bn.hip)."""
import torch

from . import _lib, ops
from .ops import once_differentiable
from ._lib import call, dtype_code, ptr, require_device, stream_ptr, scratch_key

_ws_cache = {}


def _al16(t):
    """The vectorised reduce / apply kernels issue 16-byte accesses: re-home the rare contiguous view that
    does not start on a 16-byte boundary."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _ws(device, F):
    need = int(_lib.load().cplxamd_bn_ws_bytes(F))
    key = scratch_key(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = _ws_cache[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return buf


def _geom(x):
    B, F = x.shape[0], x.shape[1]
    S = 1
    for d in x.shape[2:]:
        S *= d
    return B, F, S


def _is_cl(x):
    """4-d, stored channels-last (and not at the same time plain contiguous: C == 1 or H * W == 1)."""
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _prep(xr, xi):
    """-> (xr, xi, (B, F, S), channels_last): channels-last 4-d inputs are normalised as [B H W, F] rows (the output keeps
    the layout, so a channels-last convolution on either side needs no copy); anything else as [B, F, S] planes."""
    if _is_cl(xr) and _is_cl(xi):
        B, F, H, W = xr.shape
        return _al16(xr), _al16(xi), (B * H * W, F, 1), True
    xr, xi = _al16(xr.contiguous()), _al16(xi.contiguous())
    return xr, xi, _geom(xr), False


def _sync_group(process_group, training):
    """The process group whose ranks share batch statistics, or None: local statistics (the reference's behaviour
    per process).  `process_group` True = the default group."""
    import torch.distributed as dist
    from . import dp
    if not training or process_group is None or process_group is False or not dp.is_initialized():
        return None
    group = dist.group.WORLD if process_group is True else process_group
    return group if (dist.get_world_size(group) > 1 or dp.FORCE_COLLECTIVES) else None


def _sum_over_ranks(t, group):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class CplxBatchNormFn(torch.autograd.Function):
    """cplx_batch_norm (cplxmodule/nn/modules/batchnorm.py:189-278) incl. whiten2x2 (:62-123).

    `process_group` (None = the reference's local-batch statistics): training-mode statistics over the batches of all
    ranks of the group (optional SyncBN) -- the [F][5] forward moments and the [F][6] backward sums,
    each with the position count in the same float64 buffer, cross the ranks in ONE all-reduce per pass; the gradients
    of weight / bias stay local sums (the data-parallel exchange averages them like any other parameter)."""

    @staticmethod
    def forward(ctx, xr, xi, weight, bias, running_mean, running_var, training, momentum, eps, process_group=None,
                tracked=None):
        require_device(xr, xi, weight, bias, running_mean, running_var)
        if not training and running_mean is None:
            raise ValueError("evaluation mode requires running statistics")
        # the convolution that produced x may have left this layer's moments on it (ops.attach_moments); if it could have
        # but was not asked to, ask for the next step (conv.want_moments) -- in evaluation mode take the request back
        hint = ops.moments_hint(xr, xi) if training else None
        src = getattr(xr, "_cplxamd_conv_src", None)
        # the backward can hand its apply pass to the weight gradient of the convolution that produced x (conv.bn_fold_node)
        ctx.fold = None
        ctx.amax = False       # x came out of a float32 convolution on half pieces: its backward wants max |dX| (see backward)
        if xr.requires_grad and xr.grad_fn is not None:
            from . import conv
            ctx.fold = conv.bn_fold_node(xr, xi)
            ctx.amax = xr.grad_fn is xi.grad_fn and bool(getattr(xr.grad_fn, "x2", False))
        xr, xi, (B, F, S), ctx.cl = _prep(xr, xi)
        yr, yi = torch.empty_like(xr), torch.empty_like(xi)      # (preserve_format: channels-last stays channels-last)
        saved = torch.empty(8, F, dtype=torch.float32, device=xr.device)
        ws = _ws(xr.device, F)
        w = None if weight is None else weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        ctx.group = _sync_group(process_group, training)
        if src is not None:
            from . import conv
            conv.want_moments(src(), on=bool(training and ctx.cl and ctx.group is None))
        if hint is not None and ctx.cl and ctx.group is None and hint[0].numel() == hint[1] * F * 5:
            call("cplxamd_bn_fwd_partials", ptr(xr), ptr(xi), ptr(yr), ptr(yi), B, F, S, ptr(w), ptr(b),
                 ptr(running_mean), ptr(running_var), ptr(saved), dtype_code(xr), momentum, eps, ptr(tracked),
                 ptr(hint[0]), hint[1], ptr(ws), ws.numel(), stream_ptr())
        elif ctx.group is not None:
            m = torch.empty(F * 5 + 1, dtype=torch.float64, device=xr.device)
            m[-1] = float(B * S)
            call("cplxamd_bn_moments", ptr(xr), ptr(xi), None, None, None, B, F, S, dtype_code(xr), ptr(m), ptr(ws),
                 ws.numel(), stream_ptr())
            _sum_over_ranks(m, ctx.group)
            ctx.count = m[-1:].clone()
            call("cplxamd_bn_fwd_sync", ptr(xr), ptr(xi), ptr(yr), ptr(yi), B, F, S, ptr(w), ptr(b),
                 ptr(running_mean), ptr(running_var), ptr(saved), dtype_code(xr), momentum, eps, ptr(m),
                 ptr(ctx.count), ptr(ws), ws.numel(), stream_ptr())
        else:
            # tracked: the module's num_batches_tracked, incremented by the finalize launch (one launch less per layer)
            call("cplxamd_bn_fwd_ex", ptr(xr), ptr(xi), ptr(yr), ptr(yi), B, F, S, ptr(w), ptr(b),
                 ptr(running_mean), ptr(running_var), ptr(saved), int(training), dtype_code(xr),
                 momentum, eps, ptr(tracked), ptr(ws), ws.numel(), stream_ptr())
        ctx.save_for_backward(xr, xi, w, saved)
        ctx.training, ctx.affine = training, weight is not None
        return yr, yi

    @staticmethod
    @once_differentiable
    def backward(ctx, gr, gi):
        xr, xi, w, saved = ctx.saved_tensors
        if ctx.cl:
            fmt = torch.channels_last
            gr, gi = _al16(gr.contiguous(memory_format=fmt)), _al16(gi.contiguous(memory_format=fmt))
            B, F, S = xr.shape[0] * xr.shape[2] * xr.shape[3], xr.shape[1], 1
        else:
            gr, gi = _al16(gr.contiguous()), _al16(gi.contiguous())
            B, F, S = _geom(xr)
        dw = db = None
        if ctx.affine:
            dw = torch.empty(2, 2, F, dtype=torch.float32, device=xr.device)
            db = torch.empty(2, F, dtype=torch.float32, device=xr.device)
        ws = _ws(xr.device, F)
        # channels-last rows: the apply pass also sums dX per channel -- the bias gradient of the convolution that
        # produced x, which finds it on the gradient tensors (ops.colsum_hint) instead of reading them once more
        sums = None
        if ctx.cl and _lib.load().cplxamd_bn_rows_path(B, F, S):
            sums = torch.empty(2, F, dtype=torch.float32, device=xr.device)
        if ctx.fold is not None and ctx.group is None and sums is not None and gr.dtype == xr.dtype == torch.bfloat16:
            # the apply pass happens inside the weight-gradient launch of the convolution that produced x: sums + finalize
            # here, then that launch writes dX (the same values), its column sums and dW, which travels on dX
            from . import conv
            coef = torch.empty(F, 12, dtype=torch.float32, device=xr.device)
            call("cplxamd_bn_bwd_coef", ptr(gr), ptr(gi), ptr(xr), ptr(xi), B, F, S, ptr(w), ptr(saved), ptr(dw), ptr(db),
                 int(ctx.training), dtype_code(xr), ptr(coef), ptr(sums), ptr(ws), ws.numel(), stream_ptr())
            out = conv.cl_wgrad_bn(gr, gi, xr, xi, coef, ctx.fold)
            if out is not None:
                dxr, dxi, dwh = out
                ops.attach_colsum(dxr, sums[0])
                ops.attach_colsum(dxi, sums[1])
                ops.attach_wgrad(dxr, dxi, dwh)
                return dxr, dxi, dw, db, None, None, None, None, None, None, None
        dxr, dxi = torch.empty_like(xr), torch.empty_like(xi)
        if ctx.group is not None:
            local = torch.empty(F * 6, dtype=torch.float64, device=xr.device)
            call("cplxamd_bn_moments", ptr(xr), ptr(xi), ptr(gr), ptr(gi), ptr(saved), B, F, S, dtype_code(xr),
                 ptr(local), ptr(ws), ws.numel(), stream_ptr())
            total = local.clone()
            _sum_over_ranks(total, ctx.group)
            call("cplxamd_bn_bwd_sync", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(dxr), ptr(dxi), B, F, S, ptr(w),
                 ptr(saved), ptr(dw), ptr(db), dtype_code(xr), ptr(sums), ptr(total), ptr(local), ptr(ctx.count),
                 ptr(ws), ws.numel(), stream_ptr())
        elif sums is not None and ctx.amax and xr.dtype == torch.float32:
            # float32 planes in front of a float32 convolution: that layer's backward cuts dX into half pieces and needs
            # max |dX| first -- the apply pass, which has every value in registers, leaves it (ops.attach_scale)
            amax = torch.zeros(2048, dtype=torch.float32, device=xr.device)
            scale = torch.empty(2, dtype=torch.float32, device=xr.device)
            call("cplxamd_bn_bwd_sums_amax", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(dxr), ptr(dxi), B, F, S, ptr(w),
                 ptr(saved), ptr(dw), ptr(db), int(ctx.training), dtype_code(xr), ptr(sums), ptr(amax), ptr(ws), ws.numel(),
                 stream_ptr())
            call("cplxamd_absmax_scale_partials", ptr(amax), 2048, ptr(scale), stream_ptr())
            ops.attach_scale(dxr, dxi, scale)
        else:
            call("cplxamd_bn_bwd_sums", ptr(gr), ptr(gi), ptr(xr), ptr(xi), ptr(dxr), ptr(dxi), B, F, S,
                 ptr(w), ptr(saved), ptr(dw), ptr(db), int(ctx.training), dtype_code(xr), ptr(sums), ptr(ws),
                 ws.numel(), stream_ptr())
        if sums is not None:
            ops.attach_colsum(dxr, sums[0])
            ops.attach_colsum(dxi, sums[1])
        return dxr, dxi, dw, db, None, None, None, None, None, None, None


CplxBatchNormFn = ops.Route(CplxBatchNormFn, "cplx_batch_norm")      # float64: f64.py (the reference's algorithm in torch ops)
