from .odeint import odeint
from .adjoint import odeint_adjoint
