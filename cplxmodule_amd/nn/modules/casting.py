"""Real <-> complex layout layers (reference: cplxmodule/nn/modules/casting.py:7-150).
The interleaved converters run as single-pass kernels (csrc/layout.hip) when they copy along the
last dimension on the GPU; the others are views / one torch op, as in the reference."""

from ... import cplx
from .base import BaseRealToCplx, BaseCplxToReal


class InterleavedRealToCplx(BaseRealToCplx):
    """[..., 2d] (x_2k + i x_2k+1) -> Cplx [..., d].  `copy=False` (the reference's default) returns
    strided views; `copy=True` de-interleaves in one kernel pass."""

    def __init__(self, copy=False, dim=-1):
        super().__init__()
        self.copy, self.dim = copy, dim

    def forward(self, input):
        return cplx.from_interleaved_real(input, self.copy, self.dim)


class ConcatenatedRealToCplx(BaseRealToCplx):
    """[..., 2d] (x_k + i x_d+k) -> Cplx [..., d]."""

    def __init__(self, copy=False, dim=-1):
        super().__init__()
        self.copy, self.dim = copy, dim

    def forward(self, input):
        return cplx.from_concatenated_real(input, self.copy, self.dim)


class CplxToInterleavedReal(BaseCplxToReal):
    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return cplx.to_interleaved_real(input, True, self.dim)


class CplxToConcatenatedReal(BaseCplxToReal):
    def __init__(self, dim=-1):
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return cplx.to_concatenated_real(input, None, self.dim)


class AsTypeCplx(BaseRealToCplx):
    """x -> x + 0 i."""

    def forward(self, input):
        return cplx.Cplx(input)


class TensorToCplx(BaseRealToCplx):
    """[..., 2] -> Cplx [...]."""

    def forward(self, input):
        assert input.shape[-1] == 2
        return cplx.Cplx(input[..., 0], input[..., 1])


class CplxToTensor(BaseCplxToReal):
    """Cplx [...] -> [..., 2]."""

    def forward(self, input):
        return cplx.to_interleaved_real(input, False, -1)


class CplxModulus(BaseCplxToReal):
    """|z| (one kernel: cplxamd_modulus)."""

    def forward(self, input):
        return abs(input)


class CplxAngle(BaseCplxToReal):
    """arg z."""

    def forward(self, input):
        return input.angle
