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
        self._t = None
        self.row_order = None        # optional int32 permutation: the order in which kernels walk the rows
        self.union = None            # optional row-group union plan (build_union_plan)
        self.hub = None              # optional long-row plan (build_hub_plan)

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
        if self.union is not None:
            op.union = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.union.items()}
        return op

    def set_row_order(self, order):
        """Attach a walk order (a permutation of range(n_rows)); purely a locality hint."""
        order = torch.as_tensor(np.asarray(order), dtype=torch.int32)
        assert order.numel() == self.shape[0]
        self.row_order = order.to(self.device).contiguous()
        self._view = None
        return self

    def build_union_plan(self, rows_per_group=8, cap=30):
        """Row-group union plan (see include/ndcn_hip.h): per group of consecutive rows, the distinct columns
        they reference and, per CSR entry, its index into that list.  One-off preprocessing with torch ops on
        the operator's device; groups whose union exceeds `cap` get an empty range (the kernel gathers them
        directly).  Returns the fraction of non-zeros served from a group union (0 = no reuse worth staging)."""
        n, R = self.shape[0], int(rows_per_group)
        dev = self.device
        ng = (n + R - 1) // R
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(n, device=dev), counts)
        grp = torch.div(rows, R, rounding_mode='floor')
        key = grp * self.shape[1] + self.colidx.to(torch.int64)
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        ugrp = torch.div(uniq, self.shape[1], rounding_mode='floor')
        usize = torch.bincount(ugrp, minlength=ng)
        ok = usize <= cap                                   # groups that fit the LDS stage
        uptr_all = torch.zeros(ng + 1, dtype=torch.int64, device=dev)
        uptr_all[1:] = torch.cumsum(usize, 0)
        keep_u = ok[ugrp]
        kept_size = torch.where(ok, usize, torch.zeros_like(usize))
        ptr = torch.zeros(ng + 1, dtype=torch.int64, device=dev)
        ptr[1:] = torch.cumsum(kept_size, 0)
        cols = (uniq - ugrp * self.shape[1])[keep_u]
        lidx = inv - uptr_all[grp]                           # position of the entry's column inside its group
        lidx = torch.where(ok[grp], lidx, torch.zeros_like(lidx))
        staged_nnz = int(ok[grp].sum()) if self.nnz else 0
        self.union = {'rows': R, 'cap': int(kept_size.max()) if ng else 0, 'ptr': ptr.to(torch.int32).contiguous(),
                      'cols': cols.to(torch.int32).contiguous(),
                      'lidx': lidx.to(torch.int16).contiguous(),       # < cap <= 65535; reinterpreted as uint16
                      'loads_per_row': float(kept_size.sum() + (counts.sum() - staged_nnz)) / max(n, 1)}
        if cols.numel() == 0:
            self.union['cols'] = torch.zeros(1, dtype=torch.int32, device=dev)
        self._view = None
        return staged_nnz / max(self.nnz, 1)

    def build_hub_plan(self, H, threshold=64, seg=256):
        """Long-row plan (see include/ndcn_hip.h, struct ndcn_csr): rows with more than `threshold` entries are cut
        into segments of <= `seg` entries that a separate SpMM evaluates with one wave per segment; the fused RHS
        kernel then reads each hub's finished (A X) row as ONE entry of a second panel.  One-off numpy
        preprocessing.  Returns the number of hub rows (0 = no plan attached)."""
        rp = self.rowptr.cpu().numpy().astype(np.int64)
        deg = np.diff(rp)
        hubs = np.nonzero(deg > threshold)[0]
        self.hub = None
        self._view = None
        if hubs.size == 0:
            return 0
        dev = self.device
        ci, va = self.colidx.cpu().numpy(), self.val.cpu().numpy()
        n, n_cols = self.shape
        hub_deg = deg[hubs]
        # compact copy of the hub rows' entries, in hub order
        take = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in hubs])
        nseg_per = (hub_deg + seg - 1) // seg
        seg_len = np.concatenate([np.minimum(seg, d - seg * np.arange(k)) for d, k in zip(hub_deg, nseg_per)])
        seg_rowptr = np.zeros(seg_len.size + 1, dtype=np.int64)
        np.cumsum(seg_len, out=seg_rowptr[1:])
        cmb_rowptr = np.zeros(hubs.size + 1, dtype=np.int64)
        np.cumsum(nseg_per, out=cmb_rowptr[1:])
        # light operator: hub row h -> one entry (n_cols + h, 1.0)
        is_hub = np.zeros(n, dtype=bool)
        is_hub[hubs] = True
        lt_deg = np.where(is_hub, 1, deg)
        lt_rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lt_deg, out=lt_rowptr[1:])
        keep = np.ones(ci.size, dtype=bool)
        keep[take] = False
        lt_ci = np.empty(int(lt_rowptr[-1]), dtype=np.int32)
        lt_va = np.empty(int(lt_rowptr[-1]), dtype=np.float32)
        rows_of = np.repeat(np.arange(n), deg)
        pos_in_row = np.arange(ci.size) - rp[rows_of]
        dst = lt_rowptr[rows_of[keep]] + pos_in_row[keep]
        lt_ci[dst] = ci[keep]
        lt_va[dst] = va[keep]
        lt_ci[lt_rowptr[hubs]] = n_cols + np.arange(hubs.size)
        lt_va[lt_rowptr[hubs]] = 1.0
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(dev)
        self.hub = {
            'n': int(hubs.size), 'nseg': int(seg_len.size), 'H': int(H), 'nnz': int(take.size), 'lt_nnz': int(lt_rowptr[-1]),
            'threshold': int(threshold), 'rows': hubs,
            'seg_rowptr': t32(seg_rowptr), 'colidx': t32(ci[take]), 'val': torch.from_numpy(va[take].astype(np.float32)).to(dev),
            'cmb_rowptr': t32(cmb_rowptr), 'cmb_colidx': t32(np.arange(seg_len.size)),
            'cmb_val': torch.ones(seg_len.size, dtype=torch.float32, device=dev),
            'lt_rowptr': t32(lt_rowptr), 'lt_colidx': torch.from_numpy(lt_ci).to(dev), 'lt_val': torch.from_numpy(lt_va).to(dev),
            'Sseg': torch.empty(seg_len.size, H, dtype=torch.float32, device=dev),
            'S': torch.empty(hubs.size, H, dtype=torch.float32, device=dev),
        }
        return int(hubs.size)

    def ensure_plans(self, H):
        """One-off, lazy: attach the row-group union plan when the panel width has a kernel that uses it
        (H = 256) and the graph has enough neighbour sharing between consecutive rows for it to pay."""
        if H != 256 or self.union is not None or getattr(self, '_union_tried', False) or self.device.type != 'cuda':
            return self
        self._union_tried = True
        # Long-row plan: rows longer than the threshold leave the fused kernel.  Worth it only when such rows are the
        # exception (measured, 10^6 nodes: Barabasi-Albert m=5 36.8 -> 24.5 ms/step at threshold 32; G(n,p) with mean
        # degree 41, where a threshold of 32 moves nearly every row, 53 -> 57 ms/step): take the lowest threshold
        # that moves at most 5 % of the rows.
        if self.nnz and getattr(self, 'hub', None) is None:
            env = os.environ.get('NDCN_HUB_THRESHOLD')
            deg = (self.rowptr[1:] - self.rowptr[:-1])
            for thr in ([int(env)] if env else [32, 64, 128]):
                if thr <= 0:
                    break
                n_hub = int((deg > thr).sum())
                if n_hub == 0:
                    break
                if env or n_hub <= 0.05 * self.shape[0]:
                    self.build_hub_plan(H, thr)
                    break
        rows = int(os.environ.get('NDCN_UNION_ROWS', '8'))       # = the fused RHS kernel's row group
        cap = int(os.environ.get('NDCN_UNION_CAP', '30'))        # LDS rows per staged group
        if rows <= 0 or self.nnz == 0:
            return self
        self.build_union_plan(rows, cap)
        avg = self.nnz / max(self.shape[0], 1)
        if self.union['loads_per_row'] > 0.75 * avg:        # < 25 % fewer fetches: not worth the LDS round trip
            self.union = None
            self._view = None
        return self

    def view(self):
        """ctypes struct ndcn_csr borrowing this object's device arrays."""
        if self._view is None:
            self._view = _lib.CsrView(self.shape[0], self.shape[1], self.nnz,
                                      self.rowptr.data_ptr(), self.colidx.data_ptr() if self.nnz else None,
                                      self.val.data_ptr() if self.nnz else None,
                                      self.row_order.data_ptr() if self.row_order is not None else None,
                                      0, 0, None, None, None)
            if self.union is not None:
                u = self.union
                self._view.ug_rows, self._view.ug_cap = u['rows'], u['cap']
                self._view.ug_ptr, self._view.ug_cols = u['ptr'].data_ptr(), u['cols'].data_ptr()
                self._view.ug_lidx = u['lidx'].data_ptr()
            h = getattr(self, 'hub', None)
            if h is not None:
                v = self._view
                v.hub_n, v.hub_nseg, v.hub_H, v.hub_nnz, v.lt_nnz = h['n'], h['nseg'], h['H'], h['nnz'], h['lt_nnz']
                v.hub_seg_rowptr, v.hub_colidx, v.hub_val = h['seg_rowptr'].data_ptr(), h['colidx'].data_ptr(), h['val'].data_ptr()
                v.hub_cmb_rowptr, v.hub_cmb_colidx, v.hub_cmb_val = (h['cmb_rowptr'].data_ptr(), h['cmb_colidx'].data_ptr(),
                                                                       h['cmb_val'].data_ptr())
                v.lt_rowptr, v.lt_colidx, v.lt_val = h['lt_rowptr'].data_ptr(), h['lt_colidx'].data_ptr(), h['lt_val'].data_ptr()
                v.hub_Sseg, v.hub_S = h['Sseg'].data_ptr(), h['S'].data_ptr()
        return self._view

    def view_ref(self):
        return ctypes.byref(self.view())

    def scaled(self, alpha):
        return CsrOperator(self.rowptr, self.colidx, self.val * float(alpha), self.shape)

    def transpose(self):
        """CSR of A^T (cached): the operator of the SpMM backward, g_X = A^T g_Y."""
        if self._t is None:
            rows = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device),
                                           (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64))
            self._t = CsrOperator.from_coo(self.colidx.to(torch.int64), rows, self.val, (self.shape[1], self.shape[0]))
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
