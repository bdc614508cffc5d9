"""Real-valued variational dropout / ARD layers (cplxmodule/nn/relevance/real/{base,vd,ard}.py)
on the real GEMM + fused LRT / KL kernels."""
import torch

from .base import BaseARD, KLFusion
from .noise import noise
from ..utils.sparsity import SparsityStats
from ... import ops


class _RealGaussianMixin(KLFusion):
    _kl_kind = "real_vd"
    __sparsity_ignore__ = ("log_sigma2",)

    def _init_variational(self):
        self.log_sigma2 = torch.nn.Parameter(torch.empty(*self.weight.shape))
        self.reset_variational_parameters()

    def reset_variational_parameters(self):
        self.log_sigma2.data.fill_(-10.0)

    @property
    def log_alpha(self):
        return ops.LogAlphaFn.apply(self.log_sigma2, self.weight, None)

    @property
    def penalty(self):
        return ops.PenaltyFn.apply(self._kl_kind, self.log_sigma2, self.weight, None)

    def _penalty_reduced(self, reduction):
        total = self._kl_get((self.weight, self.log_sigma2))
        if total is None:
            total = ops.PenaltySumFn.apply(self._kl_kind, self.log_sigma2, self.weight, None)
        return total / self.log_sigma2.numel() if reduction == "mean" else total

    def relevance(self, *, threshold, **kwargs):
        with torch.no_grad():
            return ops.relevance_mask(self.weight, None, self.log_sigma2, threshold)

    def sparsity(self, *, threshold, **kwargs):
        with torch.no_grad():
            _, kept = ops.relevance_mask(self.weight, None, self.log_sigma2, threshold, count=True)
        return [(id(self.weight), self.weight.numel() - float(kept.item()))]

    def _draw_noise(self, shape, like):
        if noise.mode == "torch":
            return torch.randn(*shape, dtype=like.dtype, device=like.device), 0, 0
        if noise.mode == "tape":
            return noise.pop_tape(shape, like, False), 0, 0
        seed, offset = noise.next(like.device)
        return None, seed, offset


class LinearGaussian(_RealGaussianMixin, torch.nn.Linear):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias=bias)
        self._init_variational()

    def forward(self, input, eps=None):
        if not self.training:
            return ops.RealLinearFn.apply(input, self.weight, self.bias)
        seed = offset = 0
        if eps is None:
            eps, seed, offset = self._draw_noise((*input.shape[:-1], self.out_features), input)
        kind = self._kl_kind_for_forward()
        y, kl = ops.RealLinearLRTFn.apply(input, self.weight, self.bias, self.log_sigma2, eps,
                                          seed, offset, kind)
        if kind is not None:
            self._kl_put(kl, (self.weight, self.log_sigma2))
        return y


class LinearVD(LinearGaussian, SparsityStats, BaseARD):
    """Linear layer with variational dropout (softplus-sigmoid KL approximation)."""
    _kl_kind = "real_vd"


class LinearARD(LinearVD):
    """Linear layer with automatic relevance determination."""
    _kl_kind = "real_ard"


class BilinearGaussian(_RealGaussianMixin, torch.nn.Bilinear):
    """torch.nn.Bilinear with the local reparameterization (real/base.py:52-77)."""

    def __init__(self, in1_features, in2_features, out_features, bias=True):
        super().__init__(in1_features, in2_features, out_features, bias=bias)
        self._init_variational()

    def forward(self, input1, input2, eps=None):
        if not self.training:
            return ops.RealBilinearFn.apply(input1, input2, self.weight, self.bias, None, None, 0, 0)
        seed = offset = 0
        if eps is None:
            eps, seed, offset = self._draw_noise((*input1.shape[:-1], self.out_features), input1)
        return ops.RealBilinearFn.apply(input1, input2, self.weight, self.bias, self.log_sigma2, eps,
                                        seed, offset)


class BilinearVD(BilinearGaussian, SparsityStats, BaseARD):
    _kl_kind = "real_vd"


class BilinearARD(BilinearVD):
    _kl_kind = "real_ard"


class Conv2dGaussian(_RealGaussianMixin, torch.nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        from ... import conv
        return conv.real_conv2d_layer(self, input, eps)


class Conv2dVD(Conv2dGaussian, SparsityStats, BaseARD):
    _kl_kind = "real_vd"


class Conv2dARD(Conv2dVD):
    _kl_kind = "real_ard"


class Conv1dGaussian(_RealGaussianMixin, torch.nn.Conv1d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        from ... import conv
        return conv.real_conv1d_layer(self, input, eps)


class Conv1dVD(Conv1dGaussian, SparsityStats, BaseARD):
    _kl_kind = "real_vd"


class Conv1dARD(Conv1dVD):
    _kl_kind = "real_ard"


class Conv3dGaussian(_RealGaussianMixin, torch.nn.Conv3d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros"):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        if self.padding_mode != "zeros":
            raise ValueError(f"Only `zeros` padding mode is supported. Got `{self.padding_mode}`.")
        self._init_variational()

    def forward(self, input, eps=None):
        from ... import conv3d
        return conv3d.real_conv3d_layer(self, input, eps)


class Conv3dVD(Conv3dGaussian, SparsityStats, BaseARD):
    _kl_kind = "real_vd"


class Conv3dARD(Conv3dVD):
    _kl_kind = "real_ard"
