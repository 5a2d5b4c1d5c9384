"""Make the reference's own import lines resolve to this package.

The reference drivers do `import torchdiffeq as ode`, `from neural_dynamics import *` and
`from models import *` (heat_dynamics.py:12-14; dgnn.py:8,19,22).  `install()` registers this package's
modules under those names so the drivers run unchanged on the HIP path:

    import ndcn_amd.dropin; ndcn_amd.dropin.install()
"""
import sys


def install(models=False):
    from . import torchdiffeq, neural_dynamics
    sys.modules['torchdiffeq'] = torchdiffeq
    sys.modules['torchdiffeq._impl'] = torchdiffeq._impl
    sys.modules['neural_dynamics'] = neural_dynamics
    if models:
        from . import models as m
        sys.modules['models'] = m
