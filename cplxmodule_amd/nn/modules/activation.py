"""modReLU layers.  API counterpart of cplxmodule/nn/modules/activation.py:8-50;
the arithmetic is one fused kernel pair (csrc/layout.hip: modrelu_fwd / modrelu_bwd), including
the gradient of a learnable threshold."""
import torch

from ... import cplx
from .base import CplxToCplx


class _SoftThresholdModulus(CplxToCplx):
    """z -> max(|z| - tau, 0) * z / |z|, tau = `self.threshold` (float, or a Parameter that is
    broadcast against the input)."""

    def forward(self, input):
        return cplx.modrelu(input, self.threshold)


class CplxModReLU(_SoftThresholdModulus):
    """Fixed threshold when given a float; anything else (e.g. `None`) asks for ONE learnable
    threshold, drawn from U(0, 1/4) like the reference does."""

    def __init__(self, threshold=0.5):
        super().__init__()
        learn = not isinstance(threshold, float)
        self.threshold = torch.nn.Parameter(torch.rand(1).mul_(0.25)) if learn else threshold


class CplxAdaptiveModReLU(_SoftThresholdModulus):
    """A learnable threshold tensor of shape `dim` (default: one value), N(0, 0.02^2) at init;
    `CplxAdaptiveModReLU(d)` thresholds every feature of a [..., d] input separately,
    `CplxAdaptiveModReLU(c, 1, 1)` every channel of a [B, c, H, W] input."""

    def __init__(self, *dim):
        super().__init__()
        self.dim = tuple(dim) or (1,)
        self.threshold = torch.nn.Parameter(torch.randn(*self.dim).mul_(0.02))

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(map(str, self.dim))})"
