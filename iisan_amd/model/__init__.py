"""Drop-in counterpart of the reference's `model/` package (`Code_Uncached/model/__init__.py:1`): same class names,
constructor signatures, forward signatures and state-dict keys, with the arithmetic in libiisan_hip.so."""
from .model import *          # noqa: F401,F403
from .encoders import *       # noqa: F401,F403
from .modules import *        # noqa: F401,F403
from .versa import *          # noqa: F401,F403
