"""float32-accurate products on the 16-bit matrix pipe (the 'x3' / 'x2' modes of the float32 layers).

The reference's arithmetic is float32 (cplxmodule/cplx.py:641-646, nn/relevance/complex/base.py:43-56); the
float32 MFMA of gfx950 peaks at 157 TFLOP/s, the 16-bit ones at 2.5 PFLOP/s.  Two ways onto the fast pipe:

  'x3'  a float32 value is the exact sum of three bf16 values (csrc/split.hip); a product needs six of the nine piece
        products, each exact in the bf16 MFMA with float32 accumulation: 2^-24-level results at 1/6 of the bf16 rate.
  'x2'  x s = h0 + h1 to 2^-22 in IEEE half after a power-of-two scale s per operand (its largest magnitude into
        [2^14, 2^15)); three piece products, the scales undone in the GEMM: 2^-22 norm-wise at 1/3 of the half rate.

This module holds the host side: which products take which arithmetic, the piece layouts, and the launch sequences over
the MFMA GEMM kernels (cplxamd_cgemm_fl / cplxamd_rgemm_fl for bf16 pieces, cplxamd_cgemm_sc_fl / cplxamd_rgemm_sc_fl
for half pieces; float32 output, accumulate from the second launch on):

    forward   y  = x W^T      (N,N)  x -> [.. x1|x0] per row, W -> [.. w1 w1|w0 w0 w0] per row: K-concatenated launches
    data grad dx = g conj(W)  (N,T)  g -> pieces per row, W -> the same pieces stacked:        K-concatenated launches
    weight grad dW = g^T conj(x) (T,T)  one launch per piece pair (K is the batch: nothing to concatenate), accumulated in the
                                      float32 output -- which is small ([O, I])
Smallest terms first in every sequence.
"""
import os
import threading

import torch

from . import _lib
from ._lib import call, ptr, require_device, scratch_key, stream_ptr

SPLIT_A, SPLIT_B = 0, 1
OP_ID, OP_ABS2, OP_EXP, OP_MAX2 = 0, 1, 2, 3

# "auto": split operands from AUTO_MIN_WORK multiply-adds per product on (below that the exact float32-MFMA kernel is one
# launch against three to six and wins on latency), the arithmetic AUTO_KIND; "x2" / "x3": that arithmetic wherever the
# kernels take the shape; "exact": never.
_MODES = ("auto", "x2", "x3", "exact")
_state = threading.local()
_default_mode = os.environ.get("CPLXAMD_FP32", "auto")
if _default_mode not in _MODES:
    raise ValueError(f"CPLXAMD_FP32 must be one of {_MODES}, got {_default_mode!r}")
AUTO_MIN_WORK = 1 << 30
AUTO_KIND = os.environ.get("CPLXAMD_FP32_AUTO", "x2")        # what 'auto' runs above the threshold (x3: 2^-24 at twice the time)
if AUTO_KIND not in ("x2", "x3"):
    raise ValueError("CPLXAMD_FP32_AUTO must be 'x2' or 'x3'")


def get_fp32_mode():
    return getattr(_state, "mode", None) or _default_mode


def set_fp32_mode(mode):
    """Process default of the float32 layers' product arithmetic ('auto' | 'x2' | 'x3' | 'exact'); returns the previous one."""
    global _default_mode
    if mode not in _MODES:
        raise ValueError(f"fp32 mode must be one of {_MODES}, got {mode!r}")
    prev, _default_mode = _default_mode, mode
    return prev


class fp32_mode:
    """`with fp32_mode('x3'):` -- this thread's float32 products inside the block (forward passes; a layer's backward
    follows what its forward decided)."""

    def __init__(self, mode):
        if mode not in _MODES:
            raise ValueError(f"fp32 mode must be one of {_MODES}, got {mode!r}")
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_state, "mode", None)
        _state.mode = self.mode
        return self

    def __exit__(self, *exc):
        _state.mode = self.prev
        return False


def take(M, N, K, *tensors, mode=None):
    """Which split arithmetic the float32 product [M, K] x [N, K]^T runs on -- 'x3', 'x2' -- or None (the exact kernel):
    whole K tiles of the MFMA kernels (K % 32 == 0), 16-byte rows of every piece view in all three layouts
    (M % 8 == N % 8 == 0), float32 device tensors, and the mode (`mode`: what a layer's forward saw, handed to its backward)."""
    mode = mode or get_fp32_mode()
    if mode == "exact":
        return None
    if M <= 0 or N <= 0 or K <= 0 or (M % 8) or (N % 8) or (K % 32):
        return None
    if max(M, N, K) * 3 >= (1 << 22):          # the kernels' 32-bit per-lane tile offsets (leading dimension 3 K)
        return None
    for t in tensors:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            return None
    if mode == "auto":
        return AUTO_KIND if M * N * K >= AUTO_MIN_WORK else None
    return mode


class Pieces:
    """The 16-bit pieces of one float32 plane: `t` ([rows, n cols] side by side or [n, rows, cols] stacked), `kind`
    ('x3': bf16, three terms; 'x2': half, two terms), `scale` (x2: device float32[2] = {s, 1 / s}; the planes of one
    complex operand share it), `n` pieces per element."""
    __slots__ = ("t", "kind", "scale", "n")

    def __init__(self, t, kind, scale, n):
        self.t, self.kind, self.scale, self.n = t, kind, scale, n


_abs_ws = {}


def _absmax_ws(device):
    key = scratch_key(device)
    if key not in _abs_ws:
        _abs_ws[key] = torch.empty(int(_lib.load().cplxamd_absmax_ws_bytes()), dtype=torch.uint8, device=device)
    return _abs_ws[key]


def _dense(t, t2=None):
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError("split takes float32 matrices")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    if t2 is not None and (t2.stride() != t.stride() or t2.data_ptr() % 16):
        t, t2 = t.contiguous(), t2.contiguous()
    return t, t2


def scale_of(t, t2=None, op=OP_ID):
    """Device float32[2] = {s, 1 / s}: the power of two that puts max |op(t[, t2])| into [2^14, 2^15)."""
    require_device(t, t2)
    t, t2 = _dense(t, t2)
    rows, cols = t.shape
    out = torch.empty(2, dtype=torch.float32, device=t.device)
    call("cplxamd_absmax_scale", ptr(t), ptr(t2), t.stride(0) if rows > 1 else cols, rows, cols, int(op), ptr(out),
         ptr(_absmax_ws(t.device)), stream_ptr())
    return out


def split(t, pattern=SPLIT_A, op=OP_ID, t2=None, stacked=False, kind="x3", scale=None):
    """Pieces of op(t) for a float32 matrix t [rows, cols] (csrc/split.hip).  Side by side per row
    ([rows, npieces * cols]: K-concatenation of a K-contiguous operand) or, `stacked`, [npieces, rows, cols].
    kind 'x2': `scale` (scale_of) -- made here from this plane alone when None."""
    require_device(t, t2)
    t, t2 = _dense(t, t2)
    rows, cols = t.shape
    if kind == "x3":
        npc, dt = (3 if pattern == SPLIT_A else 6), torch.bfloat16
    else:
        npc, dt = (2 if pattern == SPLIT_A else 3), torch.float16
        if scale is None:
            scale = scale_of(t, t2, op)
    if stacked:
        out = torch.empty(npc, rows, cols, dtype=dt, device=t.device)
        ld, ps = cols, rows * cols
    else:
        out = torch.empty(rows, npc * cols, dtype=dt, device=t.device)
        ld, ps = npc * cols, cols
    lds = t.stride(0) if rows > 1 else cols
    if kind == "x3":
        call("cplxamd_split3", ptr(t), ptr(t2), lds, ptr(out), ld, ps, rows, cols, int(op), int(pattern), stream_ptr())
    else:
        call("cplxamd_split2h", ptr(t), ptr(t2), lds, ptr(out), ld, ps, rows, cols, int(op), int(pattern), ptr(scale),
             stream_ptr())
    return Pieces(out, kind, scale, npc)


def split_planes(planes, pattern=SPLIT_A, stacked=False, kind="x3", scale=None):
    """Pieces of the plane(s) of one operand -- one real plane or the (re, im) pair of a complex one, which shares a scale
    (`scale`: one a producer already formed for exactly these planes, ops.scale_hint)."""
    if kind == "x2" and scale is None:
        scale = scale_of(planes[0]) if len(planes) == 1 else scale_of(planes[0], planes[1], OP_MAX2)
    return tuple(split(p, pattern, stacked=stacked, kind=kind, scale=scale) for p in planes)


def _gemm(A, a_off, a_strides, B, b_views, b_strides, M, N, K, out, accumulate, conj_b=False, bias=None, beta=None,
          emul=None, emul_exp=False):
    """One launch on piece views: A = tuple of Pieces (views start a_off elements into each row), b_views = the B tensors."""
    from . import ops
    kind = A[0].kind
    a = [p.t.reshape(-1)[a_off:] if a_off else p.t for p in A]
    sa, sb = A[0].scale, B[0].scale
    if len(A) == 2:
        return ops.cgemm(a[0], a[1], a_strides, b_views[0], b_views[1], b_strides, M, N, K, bias=bias, conj_b=conj_b,
                         out_dtype=torch.float32, out=out, accumulate=accumulate, beta=beta, emul=emul,
                         scales=(sa, sb) if kind == "x2" else None)
    b = None if bias is None else (bias[0] if isinstance(bias, (tuple, list)) else bias)
    return ops.rgemm(a[0], a_strides, b_views[0], b_strides, M, N, K, bias=b, emul=emul, out_dtype=torch.float32, out=out,
                     emul_exp=emul_exp, accumulate=accumulate, beta=beta, scales=(sa, sb) if kind == "x2" else None)


# (offset of the A suffix in pieces, first B piece, number of K-concatenated pieces) per launch, smallest terms first:
#   x3: A [x2|x1|x0], B [w2 | w1 w1 | w0 w0 w0]      x2: A [h1|h0], B [w1 | w0 w0]
_SEQ = {"x3": ((2, 0, 1), (1, 1, 2), (0, 3, 3)), "x2": ((1, 0, 1), (0, 1, 2))}


def gemm_nn(As, Bs, M, N, K, bias=None, conj_b=False):
    """C = A B^T (+ bias): As = A-side pieces per plane [M, n K], Bs = B-side pieces per plane [N, n' K] (1 plane: real,
    2 planes: complex) -> float32 planes [M, N]."""
    out = None
    na, nb = As[0].n, Bs[0].n
    for aoff, b0, cnt in _SEQ[As[0].kind]:
        b = [p.t[:, b0 * K:(b0 + cnt) * K] for p in Bs]
        out = _gemm(As, aoff * K, (na * K, 1), Bs, b, (nb * K, 1), M, N, cnt * K, out, accumulate=out is not None,
                    conj_b=conj_b, bias=bias if out is None else None)
    return out


def gemm_nt(As, Bst, M, N, K, conj_b=False):
    """C[m, n] = sum_k A[m, k] op(B[k, n]): As [M, n K] pieces per plane, Bst = STACKED B-side pieces [n', K, N] of the
    K-major operand (a weight [O, I] read as B[n = i, k = o])."""
    out = None
    na = As[0].n
    for aoff, b0, cnt in _SEQ[As[0].kind]:
        b = [p.t[b0] for p in Bst]                # (the pieces behind b0 are contiguous with it)
        out = _gemm(As, aoff * K, (na * K, 1), Bst, b, (1, N), M, N, cnt * K, out, accumulate=out is not None, conj_b=conj_b)
    return out


# (term of A, term of B) of the piece products, smallest first; term t of a pattern-A split lives in piece n - 1 - t
_TT_ORDER = {"x3": ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)), "x2": ((0, 1), (1, 0), (0, 0))}


def gemm_tt(As, Bs, M, N, K, conj_b=False, out=None, accumulate=False, beta=None, emul=None, emul_exp=False):
    """C[m, n] = sum_k A[k, m] op(B[k, n]) (the weight gradient: K = batch) from As [K, n M], Bs [K, n N]; `out`,
    `accumulate` / `beta`, `emul` as ops.cgemm / ops.rgemm (the multiplier applies to every term, the scaled
    accumulate to the first launch only)."""
    first = True
    n = As[0].n
    for ta, tb in _TT_ORDER[As[0].kind]:
        b = [p.t[:, (n - 1 - tb) * N:] for p in Bs]
        out = _gemm(As, (n - 1 - ta) * M, (1, n * M), Bs, b, (1, n * N), M, N, K, out,
                    accumulate=accumulate if first else True, conj_b=conj_b, beta=beta if first else None, emul=emul,
                    emul_exp=emul_exp)
        first = False
    return out
