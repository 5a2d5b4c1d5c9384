"""ctypes binding of libndcn_hip.so (include/ndcn_hip.h).

The shared library is the product: there is no Python or CPU fallback behind it.  If it has not been
built, or a call is made without a ROCm device, the failure is loud (`NdcnHipError`).
"""
import ctypes
import os

import torch  # noqa: F401  -- must come first: the library binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libndcn_hip.so')

ABI_VERSION = 18
PATH_FUSED2, PATH_FUSED3, PATH_HUB, PATH_HALO, PATH_SWEEP, PATH_REC, PATH_WIDE, PATH_SMALL, PATH_EXACT32, PATH_RANGE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512

OK = 0
EINVAL, EHIP, ENONFINITE, EUNDERFLOW, EMAXSTEPS, ESTATE = -1, -2, -3, -4, -5, -6

F_RELU, F_NO_GRAPH, F_NO_CONTROL, F_PACKED, F_ACCUM = 1, 2, 4, 8, 16
RK_NONE, RK_COMBINE, RK_ERROR, RK_RK4 = 0, 1, 2, 3
M_EULER, M_MIDPOINT, M_RK4, M_DOPRI5 = 0, 1, 2, 3
METHODS = {'euler': M_EULER, 'midpoint': M_MIDPOINT, 'rk4': M_RK4, 'dopri5': M_DOPRI5}
PROF_KINDS = ('spmm', 'linear', 'rhs_fused', 'combine', 'error', 'sumsq', 'interp_fit', 'interp_eval',
              'fixed_stage', 'gather_rows', 'truth_dynamics', 'combine_bwd', 'error_bwd', 'sumsq_bwd', 'dense_bwd', 'linear_gs',
              'linear_wgrad', 'relu_bwd', 'rhs_adjoint_forward_half', 'rhs_adjoint_transposed_half')


class NdcnHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__('libndcn_hip: error %d: %s' % (code, text))
        self.code = code


class CsrView(ctypes.Structure):
    """struct ndcn_csr"""
    _fields_ = [('n_rows', ctypes.c_int64), ('n_cols', ctypes.c_int64), ('nnz', ctypes.c_int64),
                ('rowptr', ctypes.c_void_p), ('colidx', ctypes.c_void_p), ('val', ctypes.c_void_p),
                ('row_order', ctypes.c_void_p), ('tile_order', ctypes.c_void_p),
                ('rec_rows', ctypes.c_int32), ('rec_cap', ctypes.c_int32), ('rec_kib', ctypes.c_int32),
                ('rec_groups', ctypes.c_int32), ('rec', ctypes.c_void_p),
                ('hub_n', ctypes.c_int32), ('hub_nseg', ctypes.c_int32), ('hub_H', ctypes.c_int32),
                ('hub_nnz', ctypes.c_int64), ('lt_nnz', ctypes.c_int64),
                ('hub_seg_rowptr', ctypes.c_void_p), ('hub_colidx', ctypes.c_void_p), ('hub_val', ctypes.c_void_p),
                ('hub_cmb_rowptr', ctypes.c_void_p), ('hub_cmb_colidx', ctypes.c_void_p), ('hub_cmb_val', ctypes.c_void_p),
                ('lt_rowptr', ctypes.c_void_p), ('lt_colidx', ctypes.c_void_p), ('lt_val', ctypes.c_void_p),
                ('hub_Sseg', ctypes.c_void_p), ('hub_S', ctypes.c_void_p),
                ('max_row_len', ctypes.c_int32), ('symmetric', ctypes.c_int32),
                ('sweep_passes', ctypes.c_int32), ('sweep_rpw', ctypes.c_int32), ('sweep_logb', ctypes.c_int32),
                ('sweep_window', ctypes.c_int32), ('sweep_rows_per_pass', ctypes.c_int64),
                ('sweep_ent', ctypes.c_void_p), ('sweep_slab', ctypes.c_void_p), ('sweep_prog', ctypes.c_void_p),
                ('sweep_S', ctypes.c_void_p), ('sweep_eye_rowptr', ctypes.c_void_p), ('sweep_eye_colidx', ctypes.c_void_p),
                ('sweep_eye_val', ctypes.c_void_p), ('sweep_eye_rec', ctypes.c_void_p), ('sweep_eye_groups', ctypes.c_int32)]


class CsrHints(ctypes.Structure):
    """struct ndcn_csr_hints"""
    _fields_ = [('lattice_row_base', ctypes.c_int64), ('lattice_n_own', ctypes.c_int64), ('n_halo', ctypes.c_int64),
                ('row_order', ctypes.c_void_p), ('group_order', ctypes.c_void_p), ('n_group_order', ctypes.c_int64),
                ('rec_rows', ctypes.c_int32), ('rec_cap', ctypes.c_int32), ('rec_kib', ctypes.c_int32),
                ('hub_threshold', ctypes.c_int32), ('flags', ctypes.c_uint32)]


PLAN_NO_REC, PLAN_NO_STENCIL, PLAN_NO_TILE_ORDER, PLAN_NO_HUB, PLAN_EXTERNAL_SCRATCH, PLAN_ORDER_ONLY = 1, 2, 4, 8, 16, 32
PLAN_NO_SWEEP, PLAN_FORCE_SWEEP = 64, 128


def empty_csr(n_rows):
    """struct ndcn_csr of an operator that is never applied (NDCN_F_NO_GRAPH): only n_rows is read."""
    v = CsrView()
    v.n_rows = v.n_cols = int(n_rows)
    return v


class ShardBlock(ctypes.Structure):
    _fields_ = [('row_lo', ctypes.c_int64), ('row_hi', ctypes.c_int64), ('needs_halo', ctypes.c_int32), ('A', CsrView)]


class ShardView(ctypes.Structure):
    """struct ndcn_shard"""
    _fields_ = [('comm', ctypes.c_void_p), ('halo', ctypes.c_void_p), ('n_global_rows', ctypes.c_int64),
                ('n_blocks', ctypes.c_int32), ('blocks', ShardBlock * 4), ('A_own', CsrView), ('X_halo', ctypes.c_void_p)]


class SolverDesc(ctypes.Structure):
    """struct ndcn_solver_desc"""
    _fields_ = [('method', ctypes.c_int), ('H', ctypes.c_int), ('rhs_flags', ctypes.c_uint32),
                ('use_graph', ctypes.c_int), ('A', CsrView), ('W', ctypes.c_void_p), ('b', ctypes.c_void_p),
                ('rtol', ctypes.c_double), ('atol', ctypes.c_double), ('max_num_steps', ctypes.c_int64),
                ('safety', ctypes.c_double), ('ifactor', ctypes.c_double), ('dfactor', ctypes.c_double),
                ('shard', ctypes.POINTER(ShardView))]


_P, _I, _L, _F, _D, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_uint32
_CSR = ctypes.POINTER(CsrView)

# name -> (restype, argtypes); the single source of truth for "what include/ndcn_hip.h declares"
SIGNATURES = {
    'ndcn_abi_version': (_I, []),
    'ndcn_last_error': (ctypes.c_char_p, []),
    'ndcn_device_info': (_I, [ctypes.POINTER(_L)]),
    'ndcn_csr_create': (_I, [_L, _L, _L, _P, _P, _P, _I, ctypes.POINTER(CsrHints), _P, ctypes.POINTER(_P)]),
    'ndcn_csr_destroy': (_I, [_P]),
    'ndcn_csr_view': (_CSR, [_P]),
    'ndcn_csr_info': (_I, [_P, ctypes.POINTER(_L)]),
    'ndcn_csr_group_order': (_P, [_P]),
    'ndcn_csr_halo_panel': (_P, [_P]),
    'ndcn_csr_set_hub_scratch': (_I, [_P, _P, _P]),
    'ndcn_csr_sweep_info': (_I, [_P, ctypes.POINTER(_L)]),
    'ndcn_csr_set_sweep_scratch': (_I, [_P, _P]),
    'ndcn_spmm_f32': (_I, [_CSR, _P, _P, _L, _P, _I, _F, _U, _P]),
    'ndcn_linear_f32': (_I, [_P, _P, _P, _P, _L, _I, _I, _U, _P]),
    'ndcn_gcn_f32': (_I, [_CSR, _P, _P, _P, _P, _P, _I, _I, _U, _P]),
    'ndcn_gcn_work_bytes': (_L, [_L, _I]),
    'ndcn_linear_bwd_f32': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _U, _P]),
    'ndcn_linear_bwd_work_bytes': (_L, [_L, _I, _I]),
    'ndcn_scale_f32': (_I, [_P, _P, _F, _L, _P]),
    'ndcn_relu_bwd_f32': (_I, [_P, _P, _P, _L, _P]),
    'ndcn_copy_f32': (_I, [_P, _P, _L, _P]),
    'ndcn_rk_bwd_ws_bytes': (_L, []),
    'ndcn_rk_combine_bwd_f32': (_I, [_P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, ctypes.POINTER(_P), ctypes.POINTER(_P), _P, _P,
                                _P, _P, _L, _P]),
    'ndcn_solve_small_keep_supported': (_I, [_P, _I, ctypes.c_uint32]),
    'ndcn_solve_small_keep_f32': (_I, [_P, _P, _P, _I, ctypes.c_uint32, _P, ctypes.POINTER(_F), _L, _P, _P, _P]),
    'ndcn_solve_small_bwd_keep_f32': (_I, [_P, _P, _P, _P, _I, ctypes.c_uint32, _P, _P, ctypes.POINTER(_F), _L, _P, _P, _P, _P, _P]),
    'ndcn_tape_dopri5_f32': (_I, [_P, _P, _P, _P, _I, ctypes.c_uint32, _P, ctypes.POINTER(_D), _L, _D, _D, ctypes.POINTER(_D), _P, _P, _P,
                             ctypes.POINTER(_P), _P]),
    'ndcn_tape_backward_f32': (_I, [_P, _P, _P, _P, _P, _P]),
    'ndcn_tape_steplog': (_L, [_P, ctypes.POINTER(_D), _L]),
    'ndcn_tape_nfe': (_L, [_P]),
    'ndcn_tape_destroy': (None, [_P]),
    'ndcn_fixed_grid_train_f32': (_I, [_P, _P, _P, _I, ctypes.c_uint32, _I, _P, ctypes.POINTER(_F), _L, _P, _P, _P, _P]),
    'ndcn_fixed_grid_backward_f32': (_I, [_P, _P, _P, _P, _I, ctypes.c_uint32, _I, _P, _P, ctypes.POINTER(_F), _L, _P, _P, _P, _P, _P, _P]),
    'ndcn_rk_dot_diff_f32': (_I, [_P, _P, _P, _P, _P, _L, _P]),
    'ndcn_rk_error_bwd_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _F, _F, _F, _D, _P, _P,
                              ctypes.POINTER(_P), _P, _P, ctypes.POINTER(_P), _P, _P, _L, _P]),
    'ndcn_rk_rms_bwd_f32': (_I, [_P, _P, _P, _F, _F, _F, _P, _P, _P, _L, _P]),
    'ndcn_dopri5_interp_bwd_f32': (_I, [_P, _P, _P, ctypes.POINTER(_P), _F, _F, _P, _P, ctypes.POINTER(_P), _P, _P, ctypes.POINTER(_P),
                                   _P, _P, _L, _P]),
    'ndcn_dopri5_interp_direct_multi_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _F, ctypes.POINTER(_F), ctypes.POINTER(_P),
                                            _I, _L, _P]),
    'ndcn_dopri5_interp_bwd_multi_f32': (_I, [ctypes.POINTER(_P), _I, _P, _P, ctypes.POINTER(_P), _F, ctypes.POINTER(_F), _P, _P,
                                         ctypes.POINTER(_P), _P, _P, ctypes.POINTER(_P), _P, _P, _L, _P]),
    'ndcn_rhs_f32': (_I, [_CSR, _P, _P, _L, _P, _P, _P, _P, _I, _U, _P]),
    'ndcn_rhs_work_bytes': (_L, [_L, _I, _U]),
    'ndcn_adjoint_rhs_f32': (_I, [_CSR, _CSR, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _U, _P]),
    'ndcn_adjoint_rhs_work_bytes': (_L, [_L, _I, _U]),
    'ndcn_rhs_rk_f32': (_I, [_CSR, _P, _P, _L, _P, _P, _P, _P, _I, _U, _I, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I,
                        _P, _P, _P, ctypes.POINTER(_F), _F, _F, _P, _P, _P]),
    'ndcn_set_aten_norm_max': (_L, [_L]),
    'ndcn_rhs_adj_supported': (_I, [_CSR, _I, _U, _I, _I]),
    'ndcn_rhs_rk_adj_f32': (_I, [_CSR, _P, _P, _P, _P, _P, _P, _P, _I, _U, _I, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _P, _P,
                            _F, _F, _P, _P, _P]),
    'ndcn_rhs_xadd_supported': (_I, [_CSR, _I, _U, _I, _I]),
    'ndcn_rhs_rk_xadd_f32': (_I, [_CSR, _P, _P, _F, _P, _P, _P, _P, _I, _U, _P, _P, ctypes.POINTER(_F), _P, _P]),
    'ndcn_solve_small_supported': (_I, [_CSR, _I, _U, _I, _I]),
    'ndcn_solve_small_f32': (_I, [_CSR, _P, _P, _I, _U, _I, _P, ctypes.POINTER(_F), _L, _P, _P]),
    'ndcn_solve_small_bwd_f32': (_I, [_CSR, _CSR, _P, _P, _I, _U, _I, _P, _P, ctypes.POINTER(_F), _L, _P, _P, _P, _P]),
    'ndcn_gather_rows_f32': (_I, [_P, _P, _L, _I, _P, _P]),
    'ndcn_rk_combine_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _L, _P]),
    'ndcn_rk_error_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _F, _F, _L, _P, _P, _P]),
    'ndcn_scaled_sumsq_f32': (_I, [_P, _P, _P, _F, _F, _L, _P, _P, _P]),
    'ndcn_reduce_ws_bytes': (_L, []),
    'ndcn_dopri5_interp_fit_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _F, _P, _P, _P, _P, _L, _P]),
    'ndcn_dopri5_interp_direct_f32': (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _F, ctypes.POINTER(_F), _P, _L, _P]),
    'ndcn_interp_eval_f32': (_I, [_P, _P, _P, _P, _P, ctypes.POINTER(_F), _P, _L, _P]),
    'ndcn_fixed_stage_f32': (_I, [_I, _P, _P, _P, _P, _P, _P, _F, _L, _P]),
    'ndcn_row_l1_normalize_f32': (_I, [_P, _P, _L, _I, _P]),
    'ndcn_row_l1_normalize_bwd_f32': (_I, [_P, _P, _P, _L, _I, _P]),
    'ndcn_gene_rhs_f32': (_I, [_CSR, _P, _P, _F, _F, _F, _P]),
    'ndcn_mutual_rhs_f32': (_I, [_CSR, _P, _P, _F, _F, _F, _F, _F, _F, _P]),
    'ndcn_comm_unique_id': (_I, [ctypes.c_char_p]),
    'ndcn_comm_create': (_I, [ctypes.c_char_p, _I, _I, ctypes.POINTER(_P)]),
    'ndcn_comm_adopt': (_I, [_P, _I, _I, ctypes.POINTER(_P)]),
    'ndcn_comm_create_loopback': (_I, [ctypes.c_char_p, _I, _I, ctypes.POINTER(_P)]),
    'ndcn_comm_destroy': (_I, [_P]),
    'ndcn_comm_allreduce_sum_f64': (_I, [_P, _P, _I, _P]),
    'ndcn_halo_plan_create': (_I, [_P, _L, ctypes.POINTER(_L), ctypes.POINTER(_L), _P, _I, ctypes.POINTER(_P)]),
    'ndcn_halo_plan_destroy': (_I, [_P]),
    'ndcn_halo_exchange_f32': (_I, [_P, _P, _I, _P, _P, _P]),
    'ndcn_solver_workspace_bytes': (_L, [ctypes.POINTER(SolverDesc)]),
    'ndcn_solver_create': (_I, [ctypes.POINTER(SolverDesc), _P, _L, ctypes.POINTER(_P)]),
    'ndcn_solver_destroy': (_I, [_P]),
    'ndcn_solver_begin': (_I, [_P, _P, _D, _P]),
    'ndcn_solver_begin_borrowed': (_I, [_P, _P, _D, _P]),
    'ndcn_solver_advance': (_I, [_P, _D, _P, _L, _P]),
    'ndcn_solver_advance_many': (_I, [_P, ctypes.POINTER(_D), _L, _P, _P]),
    'ndcn_solver_stats': (_I, [_P, ctypes.POINTER(_D)]),
    'ndcn_solver_steplog': (_L, [_P, ctypes.POINTER(_D), _L]),
    'ndcn_prof_enable': (_I, [_I]),
    'ndcn_prof_read': (_I, [ctypes.POINTER(_D), _I]),
    'ndcn_prof_kinds': (_I, []),
    'ndcn_debug_last_rhs_path': (_I, []),
    'ndcn_set_range_guard': (_I, [_I]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises NdcnHipError when the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NdcnHipError(EHIP, '%s not built - run `python -c "import __graft_entry__ as g; g.build()"` '
                                 '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.ndcn_abi_version() != ABI_VERSION:
        raise NdcnHipError(EINVAL, 'ABI version mismatch: library %d, binding %d' % (lib.ndcn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code):
    if code < 0:
        raise NdcnHipError(code, load().ndcn_last_error().decode('utf-8', 'replace'))
    return code


def require_device(t, what='tensor'):
    """The HIP path is the only path: refuse host tensors instead of computing on the CPU."""
    if not t.is_cuda:
        raise NdcnHipError(EINVAL, '%s lives on %s; ndcn_amd computes on a ROCm device only (no CPU fallback). '
                                   'Move it with .to("cuda").' % (what, t.device))
    if t.dtype != torch.float32:
        raise NdcnHipError(EINVAL, '%s has dtype %s; the HIP path computes in float32' % (what, t.dtype))
    return t


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
