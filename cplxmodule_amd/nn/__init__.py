from .modules import *  # noqa: F401,F403
from .modules.base import CplxParameter  # noqa: F401
from . import init  # noqa: F401
from . import relevance  # noqa: F401
from . import masked  # noqa: F401
