"""Strided views into interleaved complex data (cplxmodule/utils/views.py:5-63)."""
import warnings


def fix_dim(dim, n_dim):
    """non-negative axis index, ValueError when out of range"""
    axis = dim + n_dim if dim < 0 else dim
    if not 0 <= axis < n_dim:
        raise ValueError(f"Dimension {dim} is out of range for {n_dim}.")
    return axis


def complex_view(x, dim=-1, squeeze=True):
    """(real, imag) views of a tensor holding re/im interleaved along `dim`: no copy, autograd
    flows into `x`.  A `dim` of size exactly 2 is dropped when `squeeze`; an odd size loses its
    last element (with a RuntimeWarning)."""
    dim = fix_dim(dim, x.dim())
    n = x.shape[dim]
    if n % 2:
        warnings.warn("Odd dimension size for the complex data unpacking: taking the least size "
                      "that fits.", RuntimeWarning)
    if n == 2 and squeeze:
        return x.select(dim, 0), x.select(dim, 1)
    even = x.narrow(dim, 0, n - n % 2)
    index = [slice(None)] * x.dim()
    index[dim] = slice(0, None, 2)
    real = even[tuple(index)]
    index[dim] = slice(1, None, 2)
    return real, even[tuple(index)]
