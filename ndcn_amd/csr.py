"""CSR container for graph operators handed to the HIP kernels.

The reference keeps its operator `A` as a dense torch matrix or a torch sparse COO tensor
(heat_dynamics.py:150-175; utils.py:12-23).  The kernels want CSR with int32 indices, fp32 values and
column-ascending rows; this module converts once and caches the result next to the tensor it came from.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


class CsrOperator:
    """rowptr int32[n_rows+1], colidx int32[nnz], val fp32[nnz]; all three on one device."""

    def __init__(self, rowptr, colidx, val, shape):
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.colidx = colidx.to(torch.int32).contiguous()
        self.val = val.to(torch.float32).contiguous()
        self.shape = (int(shape[0]), int(shape[1]))
        assert self.rowptr.numel() == self.shape[0] + 1
        assert self.colidx.numel() == self.val.numel()
        self._view = None
        self._handle = None          # ndcn_csr_handle* once plans were built (build_plans)
        self._t = None
        self.row_order = None        # optional int32 permutation: the order in which kernels walk the rows (caller's hint)
        self.group_order = None      # optional int32 row ids (-1 = empty slot): the order the records group the rows in
        self.rec = None              # what the group-record plan is (build_plans), or None
        self.tile_order = False      # the fused kernel walks its tiles in a lattice order
        self.stencil_stride = 0      # lattice stride the library detected (0: none)
        self.hub = None              # what the long-row plan is + its scratch panels, or None
        self.sweep = None            # what the column-sweep plan is + its scratch panel, or None

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_arrays(cls, indptr, indices, data, shape, device=None):
        mk = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt)
        op = cls(mk(indptr, torch.int32), mk(indices, torch.int32), mk(data, torch.float32), shape)
        return op.to(device) if device is not None else op

    @classmethod
    def from_scipy(cls, m, device=None):
        m = m.tocsr()
        m.sort_indices()
        return cls.from_arrays(m.indptr, m.indices, m.data, m.shape, device)

    @classmethod
    def from_coo(cls, rows, cols, vals, shape):
        """Entries -> CSR on the entries' device; duplicates are summed (torch coalesce semantics)."""
        n_rows, n_cols = int(shape[0]), int(shape[1])
        key = rows.to(torch.int64) * n_cols + cols.to(torch.int64)
        order = torch.argsort(key, stable=True)
        key, vals = key[order], vals[order].to(torch.float32)
        uniq, inv = torch.unique_consecutive(key, return_inverse=True)
        if uniq.numel() != key.numel():
            summed = torch.zeros(uniq.numel(), dtype=torch.float32, device=vals.device)
            summed.index_add_(0, inv, vals)
            key, vals = uniq, summed
        r = torch.div(key, n_cols, rounding_mode='floor')
        c = key - r * n_cols
        counts = torch.bincount(r, minlength=n_rows)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=vals.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        return cls(rowptr, c, vals, (n_rows, n_cols))

    @classmethod
    def from_torch(cls, A):
        """Dense matrix, sparse COO (the reference's `--sparse` layout) or sparse CSR torch tensor."""
        if isinstance(A, CsrOperator):
            return A
        if A.layout == torch.sparse_csr:
            return cls(A.crow_indices(), A.col_indices(), A.values(), A.shape)
        if A.is_sparse:
            idx, vals = A._indices(), A._values()
            return cls.from_coo(idx[0], idx[1], vals, A.shape)
        assert A.dim() == 2, 'operator must be a matrix'
        nz = A.nonzero()                       # row-major order, as torch_sensor_to_torch_sparse_tensor
        vals = A[nz[:, 0], nz[:, 1]]
        counts = torch.bincount(nz[:, 0], minlength=A.shape[0])
        rowptr = torch.zeros(A.shape[0] + 1, dtype=torch.int64, device=A.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        return cls(rowptr, nz[:, 1], vals, A.shape)

    # ------------------------------------------------------------------ accessors
    @property
    def device(self):
        return self.val.device

    @property
    def nnz(self):
        return int(self.val.numel())

    def to(self, device):
        device = torch.device(device)
        if device == self.val.device:
            return self
        op = CsrOperator(self.rowptr.to(device), self.colidx.to(device), self.val.to(device), self.shape)
        if self.row_order is not None:
            op.row_order = self.row_order.to(device)
        for k in ('lattice_hint', 'n_halo'):
            if hasattr(self, k):
                setattr(op, k, getattr(self, k))
        return op                    # plans are built where the operator lives (ensure_plans)

    def set_row_order(self, order):
        """Attach a walk order (a permutation of range(n_rows)); purely a locality hint."""
        order = torch.as_tensor(np.asarray(order), dtype=torch.int32)
        assert order.numel() == self.shape[0]
        self.row_order = order.to(self.device).contiguous()
        self._release()
        self._plans_tried = False
        return self

    REC_SHAPES = ((8, 32, 1), (16, 40, 2), (8, 48, 2))   # {rows per group, column-list capacity, record KiB}: spmm_rec.hip

    # ------------------------------------------------------------------ plans: a binding of ndcn_csr_create
    def _hints(self, H, rec_shape=None, hub_threshold=0, flags=0):
        h = _lib.CsrHints()
        hint = getattr(self, 'lattice_hint', None)
        if hint is not None:                                    # row block of a shard (ndcn_amd/sharding.py): (row_base, n_own)
            h.lattice_row_base, h.lattice_n_own = int(hint[0]), int(hint[1])
        h.n_halo = int(getattr(self, 'n_halo', 0))
        if self.row_order is not None:
            h.row_order = self.row_order.data_ptr()
        if self.group_order is not None:
            self.group_order = self.group_order.to(device=self.device, dtype=torch.int32).contiguous()
            h.group_order, h.n_group_order = self.group_order.data_ptr(), int(self.group_order.numel())
        if rec_shape is not None:
            h.rec_rows, h.rec_cap, h.rec_kib = (int(v) for v in rec_shape)
        h.hub_threshold = int(hub_threshold)
        h.flags = int(flags) | _lib.PLAN_EXTERNAL_SCRATCH      # the hubs' scratch comes from torch's allocator (below)
        return h

    def _release(self):
        # (only in the process that created the handle: a fork()ed child - multiprocessing's Manager server, a DataLoader worker -
        # inherits this object, and its garbage collector must not call into a HIP context that does not exist there)
        handles = ([self._handle] if getattr(self, '_handle', None) else []) + getattr(self, '_retired', [])
        if handles and getattr(self, '_handle_pid', None) == os.getpid():
            for h in handles:
                try:
                    _lib.load().ndcn_csr_destroy(h)
                except Exception:
                    pass
        self._handle = None
        self._retired = []
        self._view = None

    def _retire(self):
        """A re-plan (build_plans on an operator that has plans) keeps the old handle - and with it the device arrays that views
        handed out earlier point into (a DeviceSolver keeps its view for its lifetime) - until this operator dies."""
        if getattr(self, '_handle', None):
            self._retired = getattr(self, '_retired', []) + [self._handle]
        self._handle = None
        self._view = None

    def __del__(self):
        self._release()

    def build_plans(self, H, rec_shape=None, hub_threshold=0, flags=0):
        """ndcn_csr_create on this operator's arrays: the library builds, on the device, the plans its H = 256 kernels
        select on - lattice detection and walk orders, the group-record plan, the long-row plan (include/ndcn_hip.h;
        ndcn_amd/csrc/csr_plan.hip) - and this object keeps the handle and mirrors what the plans are:
          .rec  {'rows', 'cap', 'kib', 'groups', 'staged', 'loads_per_row'} or None
          .hub  {'n', 'nseg', 'H', 'nnz', 'lt_nnz', 'threshold', 'Sseg', 'halo_S', 'S'} or None (the scratch panels are
                torch tensors: a shard's exchange receives into the head of 'halo_S')
          .sweep {'passes', 'rows_per_wave', 'logb', 'window', 'entries', 'rows_per_pass', 'S'} or None: the column-sweep
                plan of operators without locality (struct ndcn_csr: sweep_*; csrc/spmm_sweep.hip)
          .group_order / .stencil_stride / .tile_order (bool) of a detected lattice."""
        if self.device.type != 'cuda':
            raise _lib.NdcnHipError(_lib.EINVAL, 'operator plans are built on the device; this operator lives on %s' % self.device)
        lib = _lib.load()
        self._retire()
        user_order = self.group_order is not None
        hints = self._hints(H, rec_shape, hub_threshold, flags)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.ndcn_csr_create(self.shape[0], self.shape[1], self.nnz, self.rowptr.data_ptr(),
                                           self.colidx.data_ptr() if self.nnz else None, self.val.data_ptr() if self.nnz else None,
                                           int(H), ctypes.byref(hints), _lib.stream_ptr(), ctypes.byref(handle)))
        self._handle, self._handle_pid = handle, os.getpid()
        info = (ctypes.c_int64 * 16)()
        _lib.check(lib.ndcn_csr_info(handle, info))
        (r_rows, r_cap, r_kib, r_groups, staged_nnz, staged_cols, stride, n_order, hub_n, hub_nseg, hub_thr, hub_nnz, lt_nnz,
         _n_halo, has_tile, _H) = (int(v) for v in info)
        self.rec = None
        if r_rows:
            self.rec = {'rows': r_rows, 'cap': r_cap, 'kib': r_kib, 'groups': r_groups,
                        'staged': staged_nnz / max(self.nnz, 1),
                        'loads_per_row': float(staged_cols + (self.nnz - staged_nnz)) / max(self.shape[0], 1)}
        self.stencil_stride = stride
        self.tile_order = bool(has_tile)
        if not user_order:
            self.group_order = None
            src = lib.ndcn_csr_group_order(handle)
            if src and n_order:
                self.group_order = torch.empty(n_order, dtype=torch.int32, device=self.device)
                with torch.cuda.device(self.device):
                    _lib.check(lib.ndcn_copy_f32(self.group_order.data_ptr(), src, n_order, _lib.stream_ptr()))   # a bit copy
        self.hub = None
        if hub_n:
            n_halo = int(getattr(self, 'n_halo', 0))
            Sseg = torch.empty(hub_nseg, H, dtype=torch.float32, device=self.device)
            halo_S = torch.empty(n_halo + hub_n, H, dtype=torch.float32, device=self.device)
            _lib.check(lib.ndcn_csr_set_hub_scratch(handle, Sseg.data_ptr(), halo_S.data_ptr()))
            self.hub = {'n': hub_n, 'nseg': hub_nseg, 'H': int(H), 'nnz': hub_nnz, 'lt_nnz': lt_nnz, 'threshold': hub_thr,
                        'Sseg': Sseg, 'halo_S': halo_S, 'S': halo_S[n_halo:]}
        self.sweep = None
        sw = (ctypes.c_int64 * 8)()
        _lib.check(lib.ndcn_csr_sweep_info(handle, sw))
        if int(sw[0]):
            S = torch.empty(self.shape[0], H, dtype=torch.float32, device=self.device)
            _lib.check(lib.ndcn_csr_set_sweep_scratch(handle, S.data_ptr()))
            self.sweep = {'passes': int(sw[0]), 'rows_per_wave': int(sw[1]), 'logb': int(sw[2]), 'window': int(sw[3]),
                          'entries': int(sw[4]), 'rows_per_pass': int(sw[5]), 'S': S}
        return self

    def build_rec_plan(self, rows_per_group=8, cap=32, kib=1):
        """Exactly this record shape on the current walk order (.group_order, .row_order, or consecutive rows), kept
        whatever it covers; no long-row plan.  Returns (fraction of the entries held by records, rows staged per output
        row)."""
        self.build_plans(256, rec_shape=(rows_per_group, cap, kib), flags=_lib.PLAN_NO_HUB | _lib.PLAN_NO_STENCIL | _lib.PLAN_NO_TILE_ORDER)
        self._plans_tried = True
        return self.rec['staged'], self.rec['loads_per_row']

    def detect_stencil_order(self):
        """The patch walk order of a 2-D lattice stencil in row-major node order (int32 numpy array, -1 = empty slot), or
        None - the library's detection alone (NDCN_PLAN_ORDER_ONLY); leaves this operator's plans untouched."""
        probe = CsrOperator(self.rowptr, self.colidx, self.val, self.shape)
        for k in ('lattice_hint', 'n_halo'):
            if hasattr(self, k):
                setattr(probe, k, getattr(self, k))
        probe.build_plans(256, flags=_lib.PLAN_ORDER_ONLY | _lib.PLAN_NO_HUB)
        self.stencil_stride = probe.stencil_stride
        return None if probe.group_order is None else probe.group_order.cpu().numpy()

    def ensure_plans(self, H):
        """One-off, lazy: attach the plans the H = 256 kernels use (the library decides which pay: include/ndcn_hip.h)."""
        if H != 256 or getattr(self, '_plans_tried', False) or self.device.type != 'cuda':
            return self
        self._plans_tried = True
        return self.build_plans(H)

    def view(self, need_symmetric=False):
        """struct ndcn_csr of this operator: a COPY of the handle's view when plans were built (never an alias of library-owned
        memory: this object writes max_row_len / symmetric into it, and a later re-plan must not pull it from under a solver that
        kept it), a bare view of the arrays otherwise.  need_symmetric: the caller reads ndcn_csr::symmetric (the one-launch
        reverse sweep) - found on first request: a transpose build and three comparisons that inference never pays."""
        if self._view is None:
            if getattr(self, '_handle', None):
                self._view = _lib.CsrView.from_buffer_copy(_lib.load().ndcn_csr_view(self._handle).contents)
            else:
                self._view = _lib.CsrView(self.shape[0], self.shape[1], self.nnz,
                                          self.rowptr.data_ptr(), self.colidx.data_ptr() if self.nnz else None,
                                          self.val.data_ptr() if self.nnz else None,
                                          self.row_order.data_ptr() if self.row_order is not None else None, None)
            small = self.nnz > 0 and self.shape[0] <= (1 << 16)          # (only the one-launch solves read the two facts)
            if small:
                self._view.max_row_len = int((self.rowptr[1:] - self.rowptr[:-1]).max())
        if need_symmetric and self._view.symmetric == 0 and self.nnz > 0 and self.shape[0] <= (1 << 16):
            sym = 2
            if self.shape[0] == self.shape[1]:
                t = self.transpose()
                sym = 1 if (torch.equal(t.rowptr, self.rowptr) and torch.equal(t.colidx, self.colidx) and torch.equal(t.val, self.val)) else 2
            self._view.symmetric = sym
        return self._view

    def view_ref(self, need_symmetric=False):
        return ctypes.byref(self.view(need_symmetric))

    def scaled(self, alpha):
        return CsrOperator(self.rowptr, self.colidx, self.val * float(alpha), self.shape)

    def transpose(self):
        """CSR of A^T (cached): the operator of the SpMM backward, g_X = A^T g_Y."""
        if self._t is None:
            rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device),
                                           (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64))
            self._t = CsrOperator.from_coo(self.colidx.to(torch.int64), rows, self.val, (self.shape[1], self.shape[0]))
            self._t._t = self                                  # (the transpose of the transpose)
        return self._t

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.cpu().numpy(), self.colidx.cpu().numpy(), self.rowptr.cpu().numpy()),
                             shape=self.shape)

    def to_torch_coo(self):
        rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device),
                                       (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64))
        return torch.sparse_coo_tensor(torch.stack([rows, self.colidx.to(torch.int64)]), self.val, self.shape)


def as_csr(A):
    """CSR view of whatever the caller holds as its operator; converted once per tensor.

    The reference holds `A` by reference on the module (neural_dynamics.py:14) and never mutates it, so
    the conversion is cached on the tensor object itself, keyed on its version counter."""
    if isinstance(A, CsrOperator):
        return A
    ver = getattr(A, '_version', 0)
    tag = getattr(A, '_ndcn_csr', None)
    if tag is not None and tag[0] == ver and tag[1].device == A.device:
        return tag[1]
    op = CsrOperator.from_torch(A)
    try:
        A._ndcn_csr = (ver, op)
    except Exception:       # objects that refuse attributes just reconvert
        pass
    return op
