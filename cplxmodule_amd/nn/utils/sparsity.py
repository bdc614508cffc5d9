"""How many entries of each tensor of a model the relevance layers consider dropped.

Counterpart of cplxmodule/nn/utils/sparsity.py:5-54.  Layers that can tell (`SparsityStats`)
report `(id(tensor), n_dropped)` pairs -- ids, not names, so that a shared parameter is counted
once -- and may name service tensors (e.g. `log_sigma2`) that are not model weights at all.
"""
import itertools
import warnings

_BUFFER_NOTE = "Since v2020.06 module's buffers are also accounted by `named_sparsity`."


class SparsityStats:
    """Mixin: `sparsity(**kwargs)` returns [(id(tensor), dropped count), ...] for the tensors this
    layer owns; names listed in `__sparsity_ignore__` are left out of every total."""
    __sparsity_ignore__ = ()

    def sparsity(self, **kwargs):
        raise NotImplementedError("Derived classes must implement a method to estimate sparsity.")


def _survey(module, prefix, kwargs):
    """(dropped count by tensor id, fully qualified names of the service tensors)."""
    dropped, service = {}, set()
    for path, layer in module.named_modules(prefix=prefix):
        if not isinstance(layer, SparsityStats):
            continue
        dropped.update(layer.sparsity(**kwargs))
        service.update(f"{path}.{leaf}" if path else leaf for leaf in layer.__sparsity_ignore__)
    return dropped, service


def named_sparsity(module, prefix="", **kwargs):
    """Yields `(name, (dropped, numel))` for every parameter and buffer that is not a service
    tensor; `kwargs` (e.g. `threshold=`, `hard=`) go to the layers' `sparsity` methods."""
    warnings.warn(_BUFFER_NOTE, FutureWarning)
    dropped, service = _survey(module, prefix, kwargs)
    tensors = itertools.chain(module.named_parameters(prefix=prefix), module.named_buffers(prefix=prefix))
    for name, tensor in tensors:
        if name in service:
            continue
        yield name, (dropped.get(id(tensor), 0.0), tensor.numel())


def sparsity(module, **kwargs):
    """Dropped fraction over all counted tensors (0 for a model without any)."""
    zeros = total = 0.0
    for _, (n_dropped, numel) in named_sparsity(module, **kwargs):
        zeros += n_dropped
        total += numel
    return zeros / max(total, 1)
