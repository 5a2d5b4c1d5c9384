"""The reference-side binding stub of INTEGRATION.md section B, kept as a file so that a test executes it
(tests/test_gpu_kernels.py::test_integration_stub_runs): ctypes only, no ndcn_amd import."""
# ndcn_hip_binding.py  -- lives next to neural_dynamics.py in the reference
import ctypes, torch

import os
_lib = ctypes.CDLL(os.environ.get('NDCN_HIP_LIB', 'libndcn_hip.so'))   # import torch first: the library binds to torch's HIP runtime

_i32, _i64, _p = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
class Csr(ctypes.Structure):                    # struct ndcn_csr (ABI 10): the optional plans stay zero = absent
    _fields_ = [('n_rows', _i64), ('n_cols', _i64), ('nnz', _i64), ('rowptr', _p), ('colidx', _p), ('val', _p),
                ('row_order', _p), ('tile_order', _p),
                ('rec_rows', _i32), ('rec_cap', _i32), ('rec_kib', _i32), ('rec_groups', _i32), ('rec', _p),
                ('hub_n', _i32), ('hub_nseg', _i32), ('hub_H', _i32), ('hub_nnz', _i64), ('lt_nnz', _i64),
                ('hub_seg_rowptr', _p), ('hub_colidx', _p), ('hub_val', _p),
                ('hub_cmb_rowptr', _p), ('hub_cmb_colidx', _p), ('hub_cmb_val', _p),
                ('lt_rowptr', _p), ('lt_colidx', _p), ('lt_val', _p), ('hub_Sseg', _p), ('hub_S', _p)]
assert _lib.ndcn_abi_version() == 10

def _check(rc):
    if rc < 0:
        raise RuntimeError(ctypes.c_char_p(_lib.ndcn_last_error()).value.decode())

_lib.ndcn_last_error.restype = ctypes.c_char_p
_lib.ndcn_rhs_work_bytes.restype = ctypes.c_int64
_lib.ndcn_rhs_work_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_uint32]
_lib.ndcn_rhs_f32.argtypes = [ctypes.POINTER(Csr)] + [ctypes.c_void_p] * 2 + [ctypes.c_int64] + \
                             [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]

def to_csr(A):                                  # once, at model construction (A is a torch COO tensor on the GPU)
    A = A.coalesce().to_sparse_csr()
    keep = (A.crow_indices().int(), A.col_indices().int(), A.values().float())
    return Csr(A.shape[0], A.shape[1], keep[2].numel(), *[k.data_ptr() for k in keep]), keep   # plans: ndcn_amd/csr.py

def odefunc_forward(csr, x, W, b, no_graph=False, no_control=False):
    """drop-in for the body of ODEFunc.forward (neural_dynamics.py:27-36, dropout 0)"""
    flags = 1 | (2 if no_graph else 0) | (4 if no_control else 0)       # NDCN_F_RELU | NO_GRAPH | NO_CONTROL
    y = torch.empty_like(x)
    work = torch.empty(_lib.ndcn_rhs_work_bytes(x.shape[0], x.shape[1], flags), dtype=torch.uint8, device=x.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _check(_lib.ndcn_rhs_f32(ctypes.byref(csr), x.data_ptr(), None, x.shape[0], W.data_ptr(), b.data_ptr(),
                             y.data_ptr(), work.data_ptr(), x.shape[1], flags, stream))
    return y
