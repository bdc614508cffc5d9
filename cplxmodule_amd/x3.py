"""float32-accurate products on the bf16 matrix pipe ("x3" mode of the float32 layers).

The reference's arithmetic is float32 (cplxmodule/cplx.py:641-646, nn/relevance/complex/base.py:43-56); the
float32 MFMA of gfx950 peaks at 157 TFLOP/s, the bf16 one at 2.5 PFLOP/s.  A float32 value is the exact sum of
three bf16 values (csrc/split.hip), a product needs six of the nine piece products, each exact in the bf16 MFMA with
float32 accumulation: 2^-24-level results at 1/6 of the bf16 rate.  This module holds the host side: which products
take the mode, the piece layouts, and the launch sequences over the EXISTING bf16 kernels (cplxamd_cgemm_fl /
cplxamd_rgemm_fl with float32 output and accumulate):

    forward   y  = x W^T      (N,N)  x -> [x2|x1|x0] per row, W -> [w2|w1 w1|w0 w0 w0] per row: 3 launches, K-concatenated
    data grad dx = g conj(W)  (N,T)  g -> [g2|g1|g0] per row, W -> the same six pieces stacked:  3 launches, K-concatenated
    weight grad dW = g^T conj(x) (T,T)  six launches on piece views (K is the batch: nothing to concatenate), accumulated in
                                      the float32 output -- which is small ([O, I])
Smallest terms first in every sequence.
"""
import os
import threading

import torch

from ._lib import call, ptr, require_device, stream_ptr

SPLIT_A, SPLIT_B = 0, 1
OP_ID, OP_ABS2, OP_EXP = 0, 1, 2

# "auto": x3 from AUTO_MIN_WORK multiply-adds per product on (below that the exact float32-MFMA kernel is one launch
# against six and wins on latency); "x3": wherever the bf16 kernels take the shape; "exact": never.
_MODES = ("auto", "x3", "exact")
_state = threading.local()
_default_mode = os.environ.get("CPLXAMD_FP32", "auto")
if _default_mode not in _MODES:
    raise ValueError(f"CPLXAMD_FP32 must be one of {_MODES}, got {_default_mode!r}")
AUTO_MIN_WORK = 1 << 30


def get_fp32_mode():
    return getattr(_state, "mode", None) or _default_mode


def set_fp32_mode(mode):
    """Process default of the float32 layers' product arithmetic ('auto' | 'x3' | 'exact'); returns the previous one."""
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
    """Whether the float32 product [M, K] x [N, K]^T runs on split operands: whole K tiles of the bf16 kernels
    (K % 32 == 0), 16-byte rows of every piece view in all three layouts (M % 8 == N % 8 == 0), float32 device tensors,
    and the mode (`mode`: the decision a layer's forward took, handed to its backward)."""
    mode = mode or get_fp32_mode()
    if mode == "exact":
        return False
    if M <= 0 or N <= 0 or K <= 0 or (M % 8) or (N % 8) or (K % 32):
        return False
    if max(M, N, K) * 3 >= (1 << 22):          # the kernels' 32-bit per-lane tile offsets (leading dimension 3 K)
        return False
    for t in tensors:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            return False
    return mode == "x3" or M * N * K >= AUTO_MIN_WORK


def split(t, pattern=SPLIT_A, op=OP_ID, t2=None, stacked=False):
    """bf16 pieces of op(t) for a float32 matrix t [rows, cols] (csrc/split.hip).  Side by side per row
    ([rows, npieces * cols]: K-concatenation of a K-contiguous operand) or, `stacked`, [npieces, rows, cols]."""
    require_device(t, t2)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError("x3.split takes a float32 matrix")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    if t2 is not None and (t2.stride() != t.stride() or t2.data_ptr() % 16):
        t, t2 = t.contiguous(), t2.contiguous()
    rows, cols = t.shape
    npc = 3 if pattern == SPLIT_A else 6
    if stacked:
        out = torch.empty(npc, rows, cols, dtype=torch.bfloat16, device=t.device)
        ld, ps = cols, rows * cols
    else:
        out = torch.empty(rows, npc * cols, dtype=torch.bfloat16, device=t.device)
        ld, ps = npc * cols, cols
    call("cplxamd_split3", ptr(t), ptr(t2), t.stride(0) if rows > 1 else cols, ptr(out), ld, ps, rows, cols,
         int(op), int(pattern), stream_ptr())
    return out


def _gemm(planes_a, a_strides, planes_b, b_strides, M, N, K, out, accumulate, conj_b=False, bias=None, beta=None,
          emul=None, emul_exp=False):
    from . import ops
    if len(planes_a) == 2:
        return ops.cgemm(planes_a[0], planes_a[1], a_strides, planes_b[0], planes_b[1], b_strides, M, N, K, bias=bias,
                         conj_b=conj_b, out_dtype=torch.float32, out=out, accumulate=accumulate, beta=beta, emul=emul)
    b = None if bias is None else (bias[0] if isinstance(bias, (tuple, list)) else bias)
    return ops.rgemm(planes_a[0], a_strides, planes_b[0], b_strides, M, N, K, bias=b, emul=emul,
                     out_dtype=torch.float32, out=out, emul_exp=emul_exp, accumulate=accumulate, beta=beta)


def gemm_nn(As, Bs, M, N, K, bias=None, conj_b=False):
    """C = A B^T (+ bias): As = per-plane A-side pieces [M, 3K], Bs = per-plane B-side pieces [N, 6K] (1 plane: real,
    2 planes: complex) -> float32 planes [M, N]."""
    out = None
    for aoff, boff, kk in ((2 * K, 0, K), (K, K, 2 * K), (0, 3 * K, 3 * K)):
        a = [t[:, aoff:] for t in As]
        b = [t[:, boff:boff + kk] for t in Bs]
        out = _gemm(a, (3 * K, 1), b, (6 * K, 1), M, N, kk, out, accumulate=out is not None, conj_b=conj_b,
                    bias=bias if out is None else None)
    return out


def gemm_nt(As, Bst, M, N, K, conj_b=False):
    """C[m, n] = sum_k A[m, k] op(B[k, n]): As [M, 3K] pieces per plane, Bst = STACKED B-side pieces [6, K, N] of the
    K-major operand (a weight [O, I] read as B[n = i, k = o])."""
    out = None
    for aoff, p0, kk in ((2 * K, 0, K), (K, 1, 2 * K), (0, 3, 3 * K)):
        a = [t[:, aoff:] for t in As]
        b = [t[p0] for t in Bst]                  # (the pieces behind p0 are contiguous with it)
        out = _gemm(a, (3 * K, 1), b, (1, N), M, N, kk, out, accumulate=out is not None, conj_b=conj_b)
    return out


# (term of A, term of B) of the six products, smallest first; term t of a pattern-A split lives in piece 2 - t
_TT_ORDER = ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))


def gemm_tt(As, Bs, M, N, K, conj_b=False, out=None, accumulate=False, beta=None, emul=None, emul_exp=False):
    """C[m, n] = sum_k A[k, m] op(B[k, n]) (the weight gradient: K = batch) from As [K, 3M], Bs [K, 3N]; `out`,
    `accumulate` / `beta`, `emul` as ops.cgemm / ops.rgemm (the multiplier applies to every term, the scaled
    accumulate to the first launch only)."""
    first = True
    for ta, tb in _TT_ORDER:
        a = [t[:, (2 - ta) * M:] for t in As]
        b = [t[:, (2 - tb) * N:] for t in Bs]
        out = _gemm(a, (1, 3 * M), b, (1, 3 * N), M, N, K, out, accumulate=accumulate if first else True,
                    conj_b=conj_b, beta=beta if first else None, emul=emul, emul_exp=emul_exp)
        first = False
    return out
