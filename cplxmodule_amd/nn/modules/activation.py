"""Complex activations (reference: cplxmodule/nn/modules/activation.py:8-62)."""
import torch

from ... import cplx
from .base import CplxToCplx, BaseCplxToReal


class CplxModReLU(CplxToCplx):
    """z -> (|z| - tau)_+ z / |z|; a non-float `threshold` (e.g. None) makes tau a learnable scalar
    initialised U(0, 0.25) as in the reference (activation.py:21-25)."""

    def __init__(self, threshold=0.5):
        super().__init__()
        if not isinstance(threshold, float):
            threshold = torch.nn.Parameter(torch.rand(1) * 0.25)
        self.threshold = threshold

    def forward(self, input):
        return cplx.modrelu(input, self.threshold)


class CplxAdaptiveModReLU(CplxToCplx):
    """modReLU with a learnable threshold tensor of shape `dim` (broadcast against the input),
    initialised N(0, 0.02^2) (activation.py:46-49)."""

    def __init__(self, *dim):
        super().__init__()
        self.dim = dim if dim else (1,)
        self.threshold = torch.nn.Parameter(torch.randn(*self.dim) * 0.02)

    def forward(self, input):
        return cplx.modrelu(input, self.threshold)

    def __repr__(self):
        body = repr(self.dim)[1:-1] if len(self.dim) > 1 else repr(self.dim[0])
        return f"{self.__class__.__name__}({body})"


class CplxModulus(BaseCplxToReal):
    def forward(self, input):
        return abs(input)


class CplxAngle(BaseCplxToReal):
    def forward(self, input):
        return input.angle
