"""The reference-side binding stub of INTEGRATION.md section B, kept as a file so that a test executes it
(tests/test_gpu_kernels.py::test_integration_stub_runs): ctypes only, no ndcn_amd import."""
# ndcn_hip_binding.py  -- lives next to neural_dynamics.py in the reference
import ctypes, torch

import os
_lib = ctypes.CDLL(os.environ.get('NDCN_HIP_LIB', 'libndcn_hip.so'))   # import torch first: the library binds to torch's HIP runtime
assert _lib.ndcn_abi_version() == 18

_i64, _p = ctypes.c_int64, ctypes.c_void_p

def _check(rc):
    if rc < 0:
        raise RuntimeError(ctypes.c_char_p(_lib.ndcn_last_error()).value.decode())

_lib.ndcn_last_error.restype = ctypes.c_char_p
_lib.ndcn_csr_create.argtypes = [_i64, _i64, _i64, _p, _p, _p, ctypes.c_int, _p, _p, ctypes.POINTER(_p)]
_lib.ndcn_csr_destroy.argtypes = [_p]
_lib.ndcn_csr_view.restype, _lib.ndcn_csr_view.argtypes = _p, [_p]     # const ndcn_csr *: passed on as an opaque pointer
_lib.ndcn_rhs_work_bytes.restype = _i64
_lib.ndcn_rhs_work_bytes.argtypes = [_i64, ctypes.c_int, ctypes.c_uint32]
_lib.ndcn_rhs_f32.argtypes = [_p] * 3 + [_i64] + [_p] * 4 + [ctypes.c_int, ctypes.c_uint32, _p]
_lib.ndcn_debug_last_rhs_path.restype = ctypes.c_int

class Operator:
    """once, at model construction (A is a torch COO tensor on the GPU, heat_dynamics.py:170-175): CSR arrays kept here,
    the library's handle on them - ndcn_csr_create builds the plans its kernels select on (include/ndcn_hip.h)"""
    def __init__(self, A, hidden_size):
        A = A.coalesce().to_sparse_csr()
        self.arrays = (A.crow_indices().int(), A.col_indices().int(), A.values().float())
        self.handle = _p()
        _check(_lib.ndcn_csr_create(A.shape[0], A.shape[1], self.arrays[2].numel(), *[k.data_ptr() for k in self.arrays],
                                    hidden_size, None, _p(torch.cuda.current_stream().cuda_stream), ctypes.byref(self.handle)))
        self.view = _lib.ndcn_csr_view(self.handle)
    def __del__(self):
        _lib.ndcn_csr_destroy(self.handle)

def odefunc_forward(op, x, W, b, no_graph=False, no_control=False):
    """drop-in for the body of ODEFunc.forward (neural_dynamics.py:27-36, dropout 0)"""
    flags = 1 | (2 if no_graph else 0) | (4 if no_control else 0)       # NDCN_F_RELU | NO_GRAPH | NO_CONTROL
    y = torch.empty_like(x)
    work = torch.empty(_lib.ndcn_rhs_work_bytes(x.shape[0], x.shape[1], flags), dtype=torch.uint8, device=x.device)
    stream = _p(torch.cuda.current_stream().cuda_stream)
    _check(_lib.ndcn_rhs_f32(op.view, x.data_ptr(), None, x.shape[0], W.data_ptr(), b.data_ptr(),
                             y.data_ptr(), work.data_ptr(), x.shape[1], flags, stream))
    return y

def last_rhs_path():
    """which kernel family ran (NDCN_PATH_*): 2 = rhs_fused3, the group-record kernel of the lattice plan"""
    return _lib.ndcn_debug_last_rhs_path()
