"""Complex-valued variational dropout / ARD layers.

API and arithmetic of cplxmodule/nn/relevance/complex/{base,vd,ard}.py; the forward pass runs
the complex GEMM + variance GEMM + fused noise-injection kernels, `.penalty` / `.relevance`
run the fused log-alpha kernels, and the exponential integral is evaluated on the device
(the reference goes device -> host -> scipy -> device, complex/vd.py:31-36).
"""
import torch

from .base import BaseARD, KLFusion
from .noise import noise
from ..utils.sparsity import SparsityStats
from ..modules.linear import CplxLinear, CplxBilinear
from ..modules.conv import CplxConv1d, CplxConv2d, CplxConv3d
from ... import ops, cplx


def torch_expi(x):
    """Differentiable Ei(x) (complex/vd.py:44)."""
    return ops.ExpiFn.apply(x)


ExpiFunction = ops.ExpiFn


class _CplxGaussianMixin(KLFusion):
    """log_sigma2 parameter (init -10), log_alpha, penalty, relevance for complex weights."""

    _kl_kind = "cplx_vd"
    __sparsity_ignore__ = ("log_sigma2",)

    def _init_variational(self):
        self.log_sigma2 = torch.nn.Parameter(torch.empty(*self.weight.shape))
        self.reset_variational_parameters()

    def reset_variational_parameters(self):
        self.log_sigma2.data.fill_(-10.0)

    @property
    def log_alpha(self):
        w = self.weight
        return ops.LogAlphaFn.apply(self.log_sigma2, w.real, w.imag)

    @property
    def penalty(self):
        w = self.weight
        return ops.PenaltyFn.apply(self._kl_kind, self.log_sigma2, w.real, w.imag)

    def _penalty_reduced(self, reduction):
        w = self.weight
        total = self._kl_get((w.real, w.imag, self.log_sigma2))
        if total is None:
            total = ops.PenaltySumFn.apply(self._kl_kind, self.log_sigma2, w.real, w.imag)
        return total / self.log_sigma2.numel() if reduction == "mean" else total

    def relevance(self, *, threshold, **kwargs):
        w = self.weight
        with torch.no_grad():
            return ops.relevance_mask(w.real, w.imag, self.log_sigma2, threshold)

    def sparsity(self, *, threshold, **kwargs):
        w = self.weight
        with torch.no_grad():
            _, kept = ops.relevance_mask(w.real, w.imag, self.log_sigma2, threshold, count=True)
        n_dropped = float(w.real.numel()) - float(kept.item())
        return [(id(w.real), n_dropped), (id(w.imag), n_dropped)]

    def _draw_noise(self, shape, like):
        """(eps_r, eps_i, seed, offset): tensors in 'torch' mode, Philox counters otherwise."""
        if noise.mode == "torch":
            e = cplx.randn(*shape, dtype=like.dtype, device=like.device)
            return e.real, e.imag, 0, 0
        if noise.mode == "tape":
            e = noise.pop_tape(shape, like, True)
            return e[0], e[1], 0, 0
        seed, offset = noise.next(like.device)
        return None, None, seed, offset


class CplxLinearGaussian(_CplxGaussianMixin, CplxLinear):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias=bias)
        self._init_variational()

    def forward(self, input, eps=None):
        if not self.training:
            return super().forward(input)
        w, b = self.weight, self.bias
        if eps is not None:
            er, ei, seed, offset = eps.real, eps.imag, 0, 0
        else:
            er, ei, seed, offset = self._draw_noise((*input.shape[:-1], self.out_features), input)
        br, bi = (None, None) if b is None else (b.real, b.imag)
        kind = self._kl_kind_for_forward()
        yr, yi, kl = ops.CplxLinearLRTFn.apply(input.real, input.imag, w.real, w.imag, br, bi,
                                               self.log_sigma2, er, ei, seed, offset, kind)
        if kind is not None:
            self._kl_put(kl, (w.real, w.imag, self.log_sigma2))
        return cplx.Cplx(yr, yi)


class CplxLinearVD(CplxLinearGaussian, SparsityStats, BaseARD):
    """Complex linear layer with variational dropout (exact KL via Ei)."""
    _kl_kind = "cplx_vd"


class CplxLinearARD(CplxLinearVD):
    """Complex linear layer with automatic relevance determination (softplus KL)."""
    _kl_kind = "cplx_ard"


class CplxBilinearGaussian(_CplxGaussianMixin, CplxBilinear):
    """Bilinear layer with the local reparameterization (complex/base.py:59-84): the variance is the
    real bilinear form of the squared moduli with exp(log_sigma2) as its weight."""

    def __init__(self, in1_features, in2_features, out_features, bias=True, conjugate=True):
        super().__init__(in1_features, in2_features, out_features, bias=bias, conjugate=conjugate)
        self._init_variational()

    def forward(self, input1, input2, eps=None):
        if not self.training:
            return super().forward(input1, input2)
        w, b = self.weight, self.bias
        if eps is not None:
            er, ei, seed, offset = eps.real, eps.imag, 0, 0
        else:
            er, ei, seed, offset = self._draw_noise((*input1.shape[:-1], self.out_features), input1)
        br, bi = (None, None) if b is None else (b.real, b.imag)
        yr, yi = ops.CplxBilinearFn.apply(input1.real, input1.imag, input2.real, input2.imag, w.real,
                                          w.imag, br, bi, bool(self.conjugate), self.log_sigma2, er, ei,
                                          seed, offset)
        return cplx.Cplx(yr, yi)


class CplxBilinearVD(CplxBilinearGaussian, SparsityStats, BaseARD):
    _kl_kind = "cplx_vd"


class CplxBilinearARD(CplxBilinearVD):
    _kl_kind = "cplx_ard"


class CplxConv2dGaussian(_CplxGaussianMixin, CplxConv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        if not self.training:
            return super().forward(input)
        from ... import conv
        return conv.cplx_conv2d_lrt(self, input, eps)


class CplxConv2dVD(CplxConv2dGaussian, SparsityStats, BaseARD):
    _kl_kind = "cplx_vd"


class CplxConv2dARD(CplxConv2dVD):
    _kl_kind = "cplx_ard"


class CplxConv1dGaussian(_CplxGaussianMixin, CplxConv1d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        if not self.training:
            return super().forward(input)
        from ... import conv
        return conv.cplx_conv1d_lrt(self, input, eps)


class CplxConv1dVD(CplxConv1dGaussian, SparsityStats, BaseARD):
    _kl_kind = "cplx_vd"


class CplxConv1dARD(CplxConv1dVD):
    _kl_kind = "cplx_ard"


class CplxConv3dGaussian(_CplxGaussianMixin, CplxConv3d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        if not self.training:
            return super().forward(input)
        from ... import conv3d
        return conv3d.cplx_conv3d_lrt(self, input, eps)


class CplxConv3dVD(CplxConv3dGaussian, SparsityStats, BaseARD):
    _kl_kind = "cplx_vd"


class CplxConv3dARD(CplxConv3dVD):
    _kl_kind = "cplx_ard"
