import numpy as np
import torch

DEV = "cuda"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().float().cpu().numpy()


def bf16_round(a):
    """numpy float32 -> values representable in bfloat16 (round to nearest even)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def assert_close(a, b, rtol, atol, what=""):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)
