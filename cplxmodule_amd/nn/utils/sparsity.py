"""Sparsity bookkeeping keyed by id(parameter) (cplxmodule/nn/utils/sparsity.py:5-54)."""
import warnings


class SparsityStats:
    __sparsity_ignore__ = ()

    def sparsity(self, **kwargs):
        raise NotImplementedError("Derived classes must implement a method to estimate sparsity.")


def named_sparsity(module, prefix="", **kwargs):
    warnings.warn("Since v2020.06 module's buffers are also accounted by `named_sparsity`.",
                  FutureWarning)
    dropped, service = {}, set()
    for name, mod in module.named_modules(prefix=prefix):
        if isinstance(mod, SparsityStats):
            stem = name + ("." if name else "")
            service.update(stem + k for k in mod.__sparsity_ignore__)
            dropped.update(mod.sparsity(**kwargs))
    for source in (module.named_parameters(prefix=prefix), module.named_buffers(prefix=prefix)):
        for name, tensor in source:
            if name not in service:
                yield name, (dropped.get(id(tensor), 0.0), tensor.numel())


def sparsity(module, **kwargs):
    n_zero = n_total = 0.0
    for _, (z, n) in named_sparsity(module, **kwargs):
        n_zero, n_total = n_zero + z, n_total + n
    return n_zero / max(n_total, 1)
