from .sparsity import sparsity, named_sparsity, SparsityStats  # noqa: F401
