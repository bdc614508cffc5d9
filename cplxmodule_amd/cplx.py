"""Planar complex tensors and the functional complex algebra of the hot path.

`Cplx` keeps the reference's contract (cplxmodule/cplx.py:10-376): a light container over two
real tensors of identical shape which are never copied on construction.  The GEMM-shaped
functions (`linear`, `conv2d`, `@`) and `abs` run on the HIP kernels of libcplxamd.so; cheap
pointwise arithmetic and views stay as torch plumbing on the two planes.
"""
import math
import os
from copy import deepcopy

import torch

from . import ops
from ._lib import CplxAmdError


class Cplx:
    """Complex tensor in split (real, imag) layout."""

    __slots__ = ("_re", "_im")

    def __new__(cls, real, imag=None):
        if isinstance(real, cls):
            return real
        if isinstance(real, complex):
            real, imag = torch.tensor(real.real), torch.tensor(real.imag)
        elif isinstance(real, float):
            if imag is None:
                imag = 0.0
            elif not isinstance(imag, float):
                raise TypeError("Imaginary part must be float.")
            real, imag = torch.tensor(real), torch.tensor(imag)
        elif not isinstance(real, torch.Tensor):
            raise TypeError("Real part must be torch.Tensor.")
        if imag is None:
            imag = torch.zeros_like(real)
        elif not isinstance(imag, torch.Tensor):
            raise TypeError("Imaginary part must be torch.Tensor.")
        if real.shape != imag.shape:
            raise ValueError("Real and imaginary parts have mistmatching shape.")
        obj = super().__new__(cls)
        obj._re, obj._im = real, imag
        return obj

    # -- parts and basic protocol ---------------------------------------------------------
    real = property(lambda self: self._re)
    imag = property(lambda self: self._im)
    shape = property(lambda self: self._re.shape)
    dtype = property(lambda self: self._re.dtype)
    device = property(lambda self: self._re.device)

    def _map(self, fn, *a, **k):
        return type(self)(fn(self._re, *a, **k), fn(self._im, *a, **k))

    def apply(self, f, *a, **k):
        """Apply `f` to the real and the imaginary planes independently."""
        return self._map(f, *a, **k)

    def __copy__(self):
        return type(self)(self._re, self._im)

    def __deepcopy__(self, memo):
        return type(self)(deepcopy(self._re, memo), deepcopy(self._im, memo))

    def __getitem__(self, key):
        return type(self)(self._re[key], self._im[key])

    def __setitem__(self, key, value):
        if isinstance(value, (Cplx, complex)):
            self._re[key], self._im[key] = value.real, value.imag
        else:
            self._re[key], self._im[key] = value, value

    def __iter__(self):
        return map(type(self), self._re, self._im)

    def __reversed__(self):
        return type(self)(reversed(self._re), reversed(self._im))

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"{type(self).__name__}(\n  real={self._re},\n  imag={self._im}\n)"

    def clone(self):
        return self._map(torch.clone)

    def detach(self):
        return self._map(torch.Tensor.detach)

    def requires_grad_(self, requires_grad=True):
        return type(self)(self._re.requires_grad_(requires_grad),
                          self._im.requires_grad_(requires_grad))

    @property
    def grad(self):
        re, im = self._re.grad, self._im.grad
        return None if re is None or im is None else type(self)(re, im)

    def is_complex(self):
        return True

    def dim(self):
        return self._re.dim()

    def size(self, *dim):
        return self._re.size(*dim)

    def item(self):
        return float(self._re) + 1j * float(self._im)

    # -- placement ------------------------------------------------------------------------
    def to(self, *a, **k):
        return self._map(torch.Tensor.to, *a, **k)

    def cuda(self, device=None, non_blocking=False):
        return self._map(torch.Tensor.cuda, device=device, non_blocking=non_blocking)

    def cpu(self):
        return self._map(torch.Tensor.cpu)

    @classmethod
    def from_numpy(cls, array):
        return cls(torch.from_numpy(array.real.copy()), torch.from_numpy(array.imag.copy()))

    def numpy(self):
        return self._re.numpy() + 1j * self._im.numpy()

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def _filled(cls, maker, sizes, dtype, device, requires_grad, imag_maker=None):
        re = maker(*sizes, dtype=dtype, device=device, requires_grad=requires_grad)
        im = (imag_maker or maker)(*sizes, dtype=dtype, device=device, requires_grad=requires_grad)
        return cls(re, im)

    @classmethod
    def empty(cls, *sizes, dtype=None, device=None, requires_grad=False):
        return cls._filled(torch.empty, sizes, dtype, device, requires_grad)

    @classmethod
    def zeros(cls, *sizes, dtype=None, device=None, requires_grad=False):
        return cls._filled(torch.zeros, sizes, dtype, device, requires_grad)

    @classmethod
    def ones(cls, *sizes, dtype=None, device=None, requires_grad=False):
        return cls._filled(torch.ones, sizes, dtype, device, requires_grad, torch.zeros)

    # -- shape manipulation ---------------------------------------------------------------
    def t(self):
        return self._map(torch.Tensor.t)

    def h(self):
        return self.conj.t()

    def flatten(self, start_dim=0, end_dim=-1):
        return self._map(torch.flatten, start_dim, end_dim)

    def view(self, *shape):
        shape = shape[0] if shape and isinstance(shape[0], tuple) else shape
        return self._map(torch.Tensor.view, *shape)

    def view_as(self, other):
        return self.view(*other.shape)

    def reshape(self, *shape):
        shape = shape[0] if shape and isinstance(shape[0], tuple) else shape
        return self._map(torch.Tensor.reshape, *shape)

    def squeeze(self, dim=None):
        return self._map(torch.squeeze) if dim is None else self._map(torch.squeeze, dim)

    def unsqueeze(self, dim):
        return self._map(torch.unsqueeze, dim)

    def permute(self, *dims):
        return self._map(torch.Tensor.permute, *dims)

    def transpose(self, dim0, dim1):
        return self._map(torch.transpose, dim0, dim1)

    # -- arithmetic -----------------------------------------------------------------------
    @property
    def conj(self):
        return type(self)(self._re, -self._im)

    def conjugate(self):
        return self.conj

    def __pos__(self):
        return self

    def __neg__(self):
        return type(self)(-self._re, -self._im)

    def __add__(self, other):
        if isinstance(other, (Cplx, complex)):
            return type(self)(self._re + other.real, self._im + other.imag)
        return type(self)(self._re + other, self._im)

    __radd__ = __iadd__ = __add__

    def __sub__(self, other):
        if isinstance(other, (Cplx, complex)):
            return type(self)(self._re - other.real, self._im - other.imag)
        return type(self)(self._re - other, self._im)

    __isub__ = __sub__

    def __rsub__(self, other):
        return -self + other

    def __mul__(self, other):
        if isinstance(other, Cplx) and ops.cplx_mul_ok(self._re, self._im, other._re, other._im):
            return type(self)(*ops.cplx_mul(self._re, self._im, other._re, other._im))      # one launch (float32: same bits as the reference's 6 kernels; bf16: float32 intermediates)
        if isinstance(other, (Cplx, complex)):
            return type(self)(self._re * other.real - self._im * other.imag,
                              self._im * other.real + self._re * other.imag)
        return type(self)(self._re * other, self._im * other)

    __rmul__ = __imul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Cplx) and ops.cplx_mul_ok(self._re, self._im, other._re, other._im):
            return type(self)(*ops.cplx_mul(self._re, self._im, other._re, other._im, div=True))
        if isinstance(other, (Cplx, complex)):
            other = Cplx(other) if isinstance(other, complex) else other
            den = other.real * other.real + other.imag * other.imag
            return self * (other.conj / den)
        return type(self)(self._re / other, self._im / other)

    __itruediv__ = __truediv__

    def __rtruediv__(self, other):
        den = self._re * self._re + self._im * self._im
        return (self.conj / den) * other

    def __matmul__(self, other):
        """Complex matrix product on the complex GEMM kernel (reference: 4 torch.matmul,
        cplxmodule/cplx.py:167-174)."""
        if not isinstance(other, Cplx):
            other = Cplx(other)
        return matmul(self, other)

    __imatmul__ = __matmul__

    def __rmatmul__(self, other):
        return matmul(Cplx(other), self)

    def __abs__(self):
        """Modulus via one fused kernel (reference: stack + norm, cplxmodule/cplx.py:183-192)."""
        if self._re.dtype in (torch.float32, torch.bfloat16) and self._im.dtype == self._re.dtype:
            return ops.AbsFn.apply(self._re, self._im)
        # other dtypes (float64 host-side utilities): the reference's own formulation
        return torch.norm(torch.stack([self._re, self._im], dim=0), p=2, dim=0)

    @property
    def angle(self):
        return torch.atan2(self._im, self._re)


# ------------------------------------------------------------------------------------------ #
#  functional API                                                                            #
# ------------------------------------------------------------------------------------------ #
def randn(*size, dtype=None, device=None, requires_grad=False):
    """Standard complex Gaussian noise: ONE normal draw of shape [2, *size] / sqrt(2), plane 0 ->
    real, plane 1 -> imag (same layout as cplxmodule/cplx.py:544-550, so that a recorded noise
    tape can be shared with the reference)."""
    z = torch.randn(2, *size, dtype=dtype, device=device) / math.sqrt(2)
    out = Cplx(z[0], z[1])
    return out.requires_grad_(True) if requires_grad else out


def randn_like(input, dtype=None, device=None, requires_grad=False):
    return randn(*input.size(), dtype=input.dtype if dtype is None else dtype,
                 device=input.device if device is None else device, requires_grad=requires_grad)


def phaseshift(input, phi=0.0):
    """z exp(i phi), phi in radians (cplxmodule/cplx.py:619-631)."""
    phi = torch.as_tensor(phi, dtype=input.real.dtype, device=input.real.device)
    return input * Cplx(torch.cos(phi), torch.sin(phi))


def linear(input, weight, bias=None):
    """y = x W^T + b on the complex GEMM kernel (replaces linear_naive, cplxmodule/cplx.py:634-648)."""
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr, yi = ops.CplxLinearFn.apply(input.real, input.imag, weight.real, weight.imag, br, bi)
    return Cplx(yr, yi)


# CPLXAMD_TRUE_3M=1: `linear_3m` really runs Gauss's three products (A/B, accuracy studies); default 0 -- see linear_3m
_TRUE_3M = os.environ.get("CPLXAMD_TRUE_3M", "0") == "1"


def linear_3m(input, weight, bias=None, true_3m=None):
    """Gauss's three-product form (cplxmodule/cplx.py:669-694).  On the reference's CPU path it is the FAST spelling of
    `linear` (0.64x the time); here it must not be a pessimisation: for bf16 activations the one-loop 4M MFMA kernel is
    both faster (2.32 vs 2.60 ms fwd+bwd at configs[1]: the K loop is issue / LDS bound, not MFMA bound, so dropping a
    quarter of the MFMAs buys nothing -- profiles/r04_gemm_w4_ab.txt section 8) and more accurate (no bf16-rounded operand
    sums), so by default `linear_3m` IS `linear`.  `true_3m=True` (or env CPLXAMD_TRUE_3M=1) runs the three real MFMA
    GEMMs + fused combine (operand sums rounded to bf16; float32 activations always take the exact 4M kernel)."""
    if not (_TRUE_3M if true_3m is None else true_3m):
        return linear(input, weight, bias)
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr, yi = ops.CplxLinearFn.apply(input.real, input.imag, weight.real, weight.imag, br, bi, 1)
    return Cplx(yr, yi)


# the remaining algorithm names of the reference (cplx.py:634-698) share the 4M kernel
linear_naive = linear_cat = linear


def bilinear(input1, input2, weight, bias=None, conjugate=True):
    """y = x1^H W x2 + b (conjugate=True) or x1^T W x2 + b (cplxmodule/cplx.py:1062-1087):
    one complex GEMM over the second input with the [O, I1, I2] weight read as [(O I1), I2],
    then the bilinear reduction kernel over the first input."""
    br, bi = (None, None) if bias is None else (bias.real, bias.imag)
    yr, yi = ops.CplxBilinearFn.apply(input1.real, input1.imag, input2.real, input2.imag, weight.real,
                                      weight.imag, br, bi, bool(conjugate), None, None, None, 0, 0)
    return Cplx(yr, yi)


bilinear_naive = bilinear_cat = bilinear


def matmul(u, v):
    """u[..., M, K] @ v[..., K, N]; batch dims of `v` (if any) must equal those of `u`."""
    ur, ui, vr, vi = u.real, u.imag, v.real, v.imag
    if vr.dim() == 2:
        lead, K = ur.shape[:-1], ur.shape[-1]
        # (x W^T) with W = v^T: feed v through the strided B operand, no copy
        a_r, a_i = ur.reshape(-1, K).contiguous(), ui.reshape(-1, K).contiguous()
        N = vr.shape[1]
        yr, yi = _MatmulFn.apply(a_r, a_i, vr.contiguous(), vi.contiguous())
        return Cplx(yr.view(*lead, N), yi.view(*lead, N))
    if ur.shape[:-2] != vr.shape[:-2]:
        raise CplxAmdError("batched complex matmul needs identical batch dimensions")
    batch = ur.shape[:-2]
    M, K, N = ur.shape[-2], ur.shape[-1], vr.shape[-1]
    flat = [t.reshape(-1, *t.shape[-2:]).contiguous() for t in (ur, ui, vr, vi)]
    nb = flat[0].shape[0]
    if nb > 65535:                                     # grid.z limit of the batched launch
        parts = [_BatchedMatmulFn.apply(*(t[lo:lo + 65535] for t in flat)) for lo in range(0, nb, 65535)]
        yr, yi = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    else:
        yr, yi = _BatchedMatmulFn.apply(*flat)
    return Cplx(yr.view(*batch, M, N), yi.view(*batch, M, N))


class _BatchedMatmulFn(torch.autograd.Function):
    """[Z, M, K] @ [Z, K, N] in one batched launch of the generic complex GEMM (blockIdx.z = entry)."""

    @staticmethod
    def forward(ctx, ar, ai, vr, vi):
        Z, M, K = ar.shape
        N = vr.shape[2]
        if ar.dtype != vr.dtype:
            raise CplxAmdError("matmul operands must share a dtype")
        ctx.save_for_backward(ar, ai, vr, vi)
        return ops.cgemm_batched(ar, ai, (K, 1, M * K), vr, vi, (1, N, K * N), Z, M, N, K)

    @staticmethod
    def backward(ctx, gr, gi):
        ar, ai, vr, vi = ctx.saved_tensors
        Z, M, K = ar.shape
        N = vr.shape[2]
        if torch.is_grad_enabled():       # create_graph=True: dA = G V^H, dV = A^H G through this Function itself
            T = lambda t: t.transpose(1, 2).contiguous()  # noqa: E731
            gr, gi = gr.contiguous(), gi.contiguous()
            dar, dai = _BatchedMatmulFn.apply(gr, gi, T(vr), -T(vi))
            dvr, dvi = _BatchedMatmulFn.apply(T(ar), -T(ai), gr, gi)
            return dar, dai, dvr, dvi
        gr, gi = gr.contiguous(), gi.contiguous()
        # dA[m,k] = sum_n G[m,n] conj(V[k,n]);  dV[k,n] = conj(sum_m A[m,k] conj(G[m,n]))
        dar, dai = ops.cgemm_batched(gr, gi, (N, 1, M * N), vr, vi, (N, 1, K * N), Z, M, K, N, conj_b=True)
        tr, ti = ops.cgemm_batched(ar, ai, (1, K, M * K), gr, gi, (1, N, M * N), Z, K, N, M, conj_b=True)
        return dar, dai, tr, -ti


class _MatmulFn(torch.autograd.Function):
    """[M,K] @ [K,N] (no conjugation) on the generic complex GEMM."""

    @staticmethod
    def forward(ctx, ar, ai, vr, vi):
        M, K = ar.shape
        N = vr.shape[1]
        ctx.save_for_backward(ar, ai, vr, vi)
        if ar.dtype != vr.dtype:
            raise CplxAmdError("matmul operands must share a dtype")
        return ops.cgemm(ar, ai, (K, 1), vr, vi, (1, N), M, N, K, out_dtype=ar.dtype)

    @staticmethod
    def backward(ctx, gr, gi):
        ar, ai, vr, vi = ctx.saved_tensors
        M, K = ar.shape
        N = vr.shape[1]
        if torch.is_grad_enabled():       # create_graph=True: dA = G V^H, dV = A^H G through this Function itself
            T = lambda t: t.t().contiguous()  # noqa: E731
            gr, gi = gr.contiguous(), gi.contiguous()
            dar, dai = _MatmulFn.apply(gr, gi, T(vr), -T(vi))
            dvr, dvi = _MatmulFn.apply(T(ar), -T(ai), gr, gi)
            return dar, dai, dvr, dvi
        gr, gi = gr.contiguous(), gi.contiguous()
        # dA = G conj(V)^T : dA[m,k] = sum_n G[m,n] conj(V[k,n])
        dar, dai = ops.cgemm(gr, gi, (N, 1), vr, vi, (N, 1), M, K, N, conj_b=True, out_dtype=ar.dtype)
        # dV = conj(A)^T G : dV[k,n] = sum_m conj(A[m,k]) G[m,n]; computed as conj(sum A conj(G))
        tr, ti = ops.cgemm(ar, ai, (1, K), gr, gi, (1, N), K, N, M, conj_b=True, out_dtype=ar.dtype)
        return dar, dai, tr, -ti


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
           padding_mode="zeros"):
    """Complex 2-d cross-correlation y = x * W + b (no conjugation), cplxmodule/cplx.py:770-838."""
    from . import conv  # late import: conv pulls in its own kernels
    return conv.cplx_conv2d(input, weight, bias, stride, padding, dilation, groups, padding_mode)


def conv1d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, padding_mode="zeros"):
    """Complex 1-d cross-correlation on [B, C, L] (cplxmodule/cplx.py:803-819): the 2-d kernels on a
    height-1 image."""
    from . import conv
    return conv.cplx_conv1d(input, weight, bias, stride, padding, dilation, groups, padding_mode)


def conv3d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, padding_mode="zeros"):
    """Complex 3-d cross-correlation on [B, C, D, H, W] (cplxmodule/cplx.py:841-857): the sum over
    the depth taps of 2-d correlations on the conv2d kernels (conv3d.py)."""
    from .conv3d import cplx_conv3d
    return cplx_conv3d(input, weight, bias, stride, padding, dilation, groups, padding_mode)


def symmetric_circular_padding(input, padding):
    """cplxmodule/cplx.py:701-714: circular padding ((p + 1) // 2, p // 2) per spatial dimension of a [B, C, L_1 .. L_n]
    complex tensor (`padding`: int or one entry per spatial dimension, fed to F.pad in the reference's order)."""
    import torch.nn.functional as F
    assert input.dim() > 2
    if isinstance(padding, int):
        padding = (input.dim() - 2) * [padding]
    assert isinstance(padding, (tuple, list)) and len(padding) + 2 == input.dim()
    expanded = []
    for pad in padding:
        expanded.extend(((pad + 1) // 2, pad // 2))
    return Cplx(F.pad(input.real, tuple(expanded), mode="circular"), F.pad(input.imag, tuple(expanded), mode="circular"))


def convnd(conv, input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, padding_mode="zeros"):
    """The reference's module-level operator (cplxmodule/cplx.py:770-800).  `conv` -- there the real torch functional that
    does the work (F.conv1d / 2d / 3d) -- only names the dimensionality here: the kernels behind conv1d / conv2d / conv3d
    run whichever is handed in (None: decided by input.dim())."""
    nd = input.dim() - 2
    name = getattr(conv, "__name__", "") if conv is not None else ""
    if name in ("conv1d", "conv2d", "conv3d") and int(name[4]) != nd:
        raise ValueError(f"{name} on a {input.dim()}-d input")
    fn = {1: conv1d, 2: conv2d, 3: conv3d}.get(nd)
    if fn is None:
        raise ValueError(f"convnd: {nd} spatial dimensions are not supported (1, 2 or 3)")
    return fn(input, weight, bias, stride, padding, dilation, groups, padding_mode)


def convnd_naive(conv, input, weight, stride=1, padding=0, dilation=1, groups=1):
    """cplxmodule/cplx.py:717-726 (four real convolutions; the grouped form): the same complex kernels."""
    return convnd(conv, input, weight, None, stride, padding, dilation, groups)


def convnd_quick(conv, input, weight, stride=1, padding=0, dilation=1):
    """cplxmodule/cplx.py:729-742 (two real convolutions with stacked filters, groups = 1): the same complex kernels."""
    return convnd(conv, input, weight, None, stride, padding, dilation, 1)


def convnd_3m(conv, input, weight, stride=1, padding=0, dilation=1, groups=1):
    """cplxmodule/cplx.py:745-767 (Gauss's three real convolutions; not wired into the reference's layers): routed to the
    four-product kernels, like linear_3m -- one fused K loop beats three launches plus operand sums on this chip."""
    return convnd(conv, input, weight, None, stride, padding, dilation, groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1,
                     dilation=1, padding_mode="zeros"):
    """Complex 2-d transposed convolution, weight [in, out / groups, kh, kw], no conjugation
    (cplxmodule/cplx.py:860-1001; the reference's functional default `groups=0` is not kept).  Runs
    the data-gradient kernels of conv2d as the forward pass."""
    from . import conv
    return conv.cplx_conv_transpose2d(input, weight, bias, stride, padding, output_padding, groups,
                                      dilation, padding_mode)


def conv_transpose1d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1,
                     dilation=1, padding_mode="zeros"):
    from . import conv
    return conv.cplx_conv_transpose1d(input, weight, bias, stride, padding, output_padding, groups,
                                      dilation, padding_mode)


def from_interleaved_real(input, copy=True, dim=-1):
    """[..., 2D] interleaved (re, im) -> Cplx [..., D]  (cplxmodule/cplx.py:451-455).  copy=True along
    the last dim on the GPU is one de-interleaving kernel pass (csrc/layout.hip) instead of two
    strided copies; copy=False returns strided views like the reference."""
    dim = dim % input.dim()
    if copy and dim == input.dim() - 1 and input.is_cuda and input.dtype in (torch.float32, torch.bfloat16):
        return Cplx(*ops.DeinterleaveFn.apply(input))
    shape = list(input.shape)
    shape[dim:dim + 1] = [shape[dim] // 2, 2]
    pair = input.reshape(shape)
    re, im = pair.select(dim + 1, 0), pair.select(dim + 1, 1)
    return Cplx(re.clone(), im.clone()) if copy else Cplx(re, im)


from_real = from_interleaved_real


def to_interleaved_real(input, flatten=True, dim=-1):
    """Cplx [..., D] -> real [..., 2D] interleaved (cplxmodule/cplx.py:466-470); last dim on the GPU:
    one interleaving kernel pass."""
    d = dim % input.dim()
    re, im = input.real, input.imag
    if d == input.dim() - 1 and re.is_cuda and re.dtype in (torch.float32, torch.bfloat16):
        out = ops.InterleaveFn.apply(re, im)
        return out if flatten else out.view(*re.shape, 2)
    out = torch.stack([re, im], dim=d + 1)
    return out.flatten(d, d + 1) if flatten else out


to_real = to_interleaved_real


def from_concatenated_real(input, copy=True, dim=-1):
    re, im = torch.chunk(input, 2, dim=dim)
    return Cplx(re.clone(), im.clone()) if copy else Cplx(re, im)


def to_concatenated_real(input, flatten=None, dim=-1):
    return torch.cat([input.real, input.imag], dim=dim)


def modrelu(input, threshold=0.5):
    """Soft-threshold of the modulus, phase kept: z * relu(1 - threshold / max(|z|, 1e-5))
    (cplxmodule/cplx.py:565-616; note the reference's flipped sign convention).  `threshold` is a
    float or a (learnable) tensor broadcastable to the input."""
    return Cplx(*ops.ModReluFn.apply(input.real, input.imag, threshold))


def dropout(input, p=0.5, training=True):
    """Complex dropout: real and imaginary parts of an element are dropped together
    (cplxmodule/nn/modules/extra.py:7-25); the Bernoulli stream is the package's Philox stream."""
    if not training or p == 0.0:
        return input
    from .nn.relevance.noise import noise
    seed, offset = noise.next(input.real.device)
    return Cplx(*ops.CplxDropoutFn.apply(input.real, input.imag, p, seed, offset))


def _pair2(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def max_pool2d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """Complex max pooling on [B, C, H, W]: the element of largest modulus in each window keeps
    both its parts (cplxmodule/cplx.py:1114-1175, 1183-1190)."""
    k = _pair2(kernel_size)
    s = k if stride is None else _pair2(stride)
    return Cplx(*ops.CplxMaxPool2dFn.apply(input.real, input.imag, k, s, _pair2(padding),
                                           _pair2(dilation), bool(ceil_mode)))


def max_pool1d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """[B, C, L] (cplxmodule/cplx.py:1178-1181): the 2-d kernel on a height-1 image."""
    first = lambda v: v if isinstance(v, int) else v[0]  # noqa: E731
    k = first(kernel_size)
    s = k if stride is None else first(stride)
    z = Cplx(input.real.unsqueeze(2), input.imag.unsqueeze(2))
    out = max_pool2d(z, (1, k), (1, s), (0, first(padding)), (1, first(dilation)), ceil_mode)
    return Cplx(out.real.squeeze(2), out.imag.squeeze(2))


def max_pool3d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False):
    """[B, C, D, H, W] (cplxmodule/cplx.py:1193-1200): separable, 2-d pool then 1-d pool over depth."""
    from .conv3d import cplx_max_pool3d
    return cplx_max_pool3d(input, kernel_size, stride, padding, dilation, ceil_mode)


# float64: a parity mode on its own kernels (f64.py)
_BatchedMatmulFn = ops.Route(_BatchedMatmulFn, "matmul_batched")
_MatmulFn = ops.Route(_MatmulFn, "matmul2d")
