"""ndcn_amd - MI355X-native implementation of the NDCN ODEFunc hot path (calvin-zcx/ndcn):
CSR SpMM graph convolution + Linear + ReLU under a dopri5 / RK4 / Euler integrator, as hand-written HIP
kernels behind the reference's own Python API.

    from ndcn_amd import torchdiffeq as ode              # odeint, odeint_adjoint
    from ndcn_amd.neural_dynamics import ODEFunc, ODEBlock, ODEBlock2, NDCN, GraphConvolution
    import ndcn_amd.dropin; ndcn_amd.dropin.install()    # makes `import torchdiffeq` / `neural_dynamics` resolve here
"""
from . import _lib
from .csr import CsrOperator, as_csr
from .ops import HipOps, hip, device_info, invalidate_packed_weights

__all__ = ['CsrOperator', 'as_csr', 'HipOps', 'hip', 'device_info', 'invalidate_packed_weights', '_lib']
