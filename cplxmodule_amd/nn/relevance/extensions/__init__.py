"""Alternative complex variational-dropout penalties (reference:
cplxmodule/nn/relevance/extensions/complex.py:18-163).  Same layers and forward pass as
CplxLinearVD / CplxConv2dVD; only the KL kind evaluated by csrc/kl.hip differs:
  *VDApprox     softplus(-la) + 0.57810 sigmoid(1.36526 (-la) - 1.45926)            (:113-117)
  *VDScaleFree  log|w| - log_sigma2 - Ei(-1/alpha) / 2  (scale-free log-uniform prior) (:43-46)
"""
from ..complex import CplxLinearVD, CplxConv2dVD, CplxBilinearVD


class CplxLinearVDApprox(CplxLinearVD):
    _kl_kind = "cplx_vd_approx"


class CplxConv2dVDApprox(CplxConv2dVD):
    _kl_kind = "cplx_vd_approx"


class CplxLinearVDScaleFree(CplxLinearVD):
    _kl_kind = "cplx_vd_scalefree"


class CplxConv2dVDScaleFree(CplxConv2dVD):
    _kl_kind = "cplx_vd_scalefree"


class CplxBilinearVDApprox(CplxBilinearVD):
    _kl_kind = "cplx_vd_approx"


class CplxBilinearVDScaleFree(CplxBilinearVD):
    _kl_kind = "cplx_vd_scalefree"
