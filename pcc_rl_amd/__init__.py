"""Import shim: the package sources live in ``pcc-rl_amd/`` (the name the build contract
prescribes, which is not a valid Python identifier); this module makes them importable as
``pcc_rl_amd``."""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pcc-rl_amd")
__path__.insert(0, _SRC)  # submodules resolve inside pcc-rl_amd/

from ._api import *  # noqa: E402,F401,F403
from ._api import __all__  # noqa: E402,F401
