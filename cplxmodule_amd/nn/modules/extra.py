"""CplxDropout (reference: cplxmodule/nn/modules/extra.py:7-25)."""
from ... import cplx
from .base import CplxToCplx


class CplxDropout(CplxToCplx):
    """Drops complex elements (real and imaginary part together) with probability p and rescales
    the survivors by 1 / (1 - p).  The mask comes from the package's counter-based Philox stream
    (`nn.relevance.noise`), is never stored, and is regenerated in backward."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        if p < 0 or p > 1:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {p}")
        self.p, self.inplace = p, inplace

    def forward(self, input):
        if self.p == 1.0 and self.training:
            return input * 0.0
        return cplx.dropout(input, self.p, self.training)

    def extra_repr(self):
        return f"p={self.p}"
