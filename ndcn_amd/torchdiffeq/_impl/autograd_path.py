"""odeint when a gradient is required: backpropagation THROUGH the solver, as the reference drivers train
(heat_dynamics.py:333, dgnn.py:204) - the reference's control flow (core.py) over differentiable HIP ops."""
from ...autograd_ops import autograd_ops
from . import core


def odeint_with_grad(func, y0, t, rtol, atol, method, options, autonomous=False):
    if method == 'dopri5':
        return core.integrate_dopri5(autograd_ops, func, y0, t, rtol, atol, autonomous=autonomous, **options)
    if options:
        raise NotImplementedError('fixed-grid options %s: only the default grid (grid == t) is provided' % sorted(options))
    return core.integrate_fixed(autograd_ops, func, y0, t, method, autonomous=autonomous)
