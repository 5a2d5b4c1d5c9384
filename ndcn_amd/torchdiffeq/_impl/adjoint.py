"""`odeint_adjoint` - O(1)-memory backward (reference: torchdiffeq/_impl/adjoint.py:105-133).

SURVEY.md 8(f) rank 1, a "next" row: the reference parses `--adjoint` but never enables it
(heat_dynamics.py:43; neural_dynamics.py:145-147).  The signature and argument checks are mirrored; the
backward itself is not built yet and says so loudly.
"""
import torch.nn as nn


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None):
    # adjoint.py:109-110
    if not isinstance(func, nn.Module):
        raise ValueError('func is required to be an instance of nn.Module.')
    raise NotImplementedError('odeint_adjoint: the adjoint backward is not part of this build yet '
                              '(SURVEY.md 8f rank 1); use odeint, which backpropagates through the solver')
