from .base import BaseARD, penalties, named_penalties, named_relevance, compute_ard_masks  # noqa: F401
from .noise import noise  # noqa: F401
from .real import BilinearVD, BilinearARD, Conv3dVD, Conv3dARD  # noqa: F401
from .real import LinearVD, LinearARD, Conv1dVD, Conv1dARD, Conv2dVD, Conv2dARD  # noqa: F401
from .complex import CplxLinearVD, CplxLinearARD, CplxConv2dVD, CplxConv2dARD  # noqa: F401
from .complex import CplxConv1dVD, CplxConv1dARD  # noqa: F401
from .complex import CplxBilinearVD, CplxBilinearARD, CplxConv3dVD, CplxConv3dARD  # noqa: F401
from .complex import torch_expi  # noqa: F401
from . import extensions  # noqa: F401
