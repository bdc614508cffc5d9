"""Alternative complex variational-dropout penalties (reference:
cplxmodule/nn/relevance/extensions/complex.py:18-206).  Same layers and forward pass as the
Cplx*VD layers; only the KL kind evaluated by csrc/kl.hip differs:
  *VDApprox     softplus(-la) + 0.57810 sigmoid(1.36526 (-la) - 1.45926)              (:113-117)
  *VDScaleFree  log|w| - log_sigma2 - Ei(-1/alpha) / 2  (scale-free log-uniform prior) (:43-46)
  *VDBogus      -log_alpha as the value, the exact KL's gradient                       (:142-160)
                (the reference's way around its host-side Ei; here the exact penalty costs the same)
"""
from .. import complex as _base

_KINDS = {"Approx": "cplx_vd_approx", "ScaleFree": "cplx_vd_scalefree", "Bogus": "cplx_vd_bogus"}
__all__ = []

for _layer in ("Linear", "Bilinear", "Conv1d", "Conv2d", "Conv3d"):
    _parent = getattr(_base, f"Cplx{_layer}VD")
    for _suffix, _kind in _KINDS.items():
        _name = f"Cplx{_layer}VD{_suffix}"
        globals()[_name] = type(_name, (_parent,), {"_kl_kind": _kind, "__module__": __name__,
                                                    "__doc__": f"{_parent.__name__} with the `{_kind}` penalty."})
        __all__.append(_name)
