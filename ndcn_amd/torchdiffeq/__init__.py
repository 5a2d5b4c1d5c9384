"""Drop-in for the reference's vendored `torchdiffeq` package (torchdiffeq/__init__.py:1-2)."""
from ._impl import odeint
from ._impl import odeint_adjoint
