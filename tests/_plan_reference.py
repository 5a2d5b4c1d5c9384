"""The plan builders of ndcn_amd/csr.py as they stood up to ABI 10 (torch / numpy, any device), kept as TEST
INFRASTRUCTURE: the product builds its plans inside libndcn_hip.so (ndcn_csr_create, ndcn_amd/csrc/csr_plan.hip) and
tests/test_gpu_plans.py compares those with this restatement bit for bit; tests/test_graphs.py checks the restatement
itself (records decode back to the operator) on the CPU.  Nothing under ndcn_amd/ imports this module."""
import os

import numpy as np
import torch


class PlanReference:
    """Plans of one operator: rowptr / colidx / val as torch tensors (int32, int32, fp32), shape (n_rows, n_cols)."""

    def __init__(self, rowptr, colidx, val, shape, lattice_hint=None, n_halo=0):
        self.rowptr, self.colidx, self.val = rowptr.to(torch.int32), colidx.to(torch.int32), val.to(torch.float32)
        self.shape = (int(shape[0]), int(shape[1]))
        self.row_order = self.group_order = self.tile_order = self.rec = self.hub = None
        self.stencil_stride = 0
        if lattice_hint is not None:
            self.lattice_hint = lattice_hint
        self.n_halo = n_halo
        self._view = None

    @classmethod
    def of(cls, op, **kw):
        """from a CsrOperator (or anything with rowptr / colidx / val / shape)"""
        return cls(op.rowptr, op.colidx, op.val, op.shape, **kw)

    @property
    def device(self):
        return self.val.device

    @property
    def nnz(self):
        return int(self.val.numel())

    REC_SHAPES = ((8, 32, 1), (16, 40, 2), (8, 48, 2))   # {rows per group, column-list capacity, record KiB}: spmm_rec.hip

    def build_rec_plan(self, rows_per_group=8, cap=32, kib=1):
        """Group-record plan (include/ndcn_hip.h, struct ndcn_csr::rec): per group of rows that are consecutive in the
        walk order (self.group_order - row ids with -1 padding -, self.row_order or 0..n-1) one fixed-size record = the group's distinct columns + per-row headers +
        the rows' entries re-indexed into that column list.  One-off preprocessing with torch ops on the operator's
        device, O(nnz log nnz).  Groups the record cannot hold (more than `cap` distinct columns, more entries than fit,
        a row longer than 64) are flagged and gathered directly by the kernel.
        Returns (fraction of non-zeros served from a staged group, rows staged into LDS per output row)."""
        n, n_cols = self.shape
        R, CAP, words = int(rows_per_group), int(cap), int(kib) * 256
        E0 = CAP + 2 * R
        ecap = (words - E0) // 2
        dev = self.device
        i64 = torch.int64
        if self.group_order is not None:                               # row ids, -1 = empty slot
            order = self.group_order.to(i64)
        elif self.row_order is not None:
            order = self.row_order.to(i64)
        else:
            order = torch.arange(n, device=dev)
        M = int(order.numel())
        ng = (M + R - 1) // R
        valid = order >= 0
        pos = torch.empty(n, dtype=i64, device=dev)
        pos[order[valid]] = torch.arange(M, device=dev)[valid]         # position of a row in the walk
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(i64)
        rows = torch.repeat_interleave(torch.arange(n, device=dev), counts)
        e_pos = pos[rows]
        e_grp = torch.div(e_pos, R, rounding_mode='floor')
        e_i = e_pos - e_grp * R
        # distinct columns per group, ascending
        key = e_grp * n_cols + self.colidx.to(i64)
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        ugrp = torch.div(uniq, n_cols, rounding_mode='floor')
        ucol = uniq - ugrp * n_cols
        usize = torch.bincount(ugrp, minlength=ng)
        ustart = torch.zeros(ng + 1, dtype=i64, device=dev)
        ustart[1:] = torch.cumsum(usize, 0)
        slot = inv - ustart[e_grp]
        # per (group, row-in-group): row id, entry count, offset of its entries inside the group
        grow = torch.full((ng * R,), -1, dtype=i64, device=dev)
        grow[pos] = torch.arange(n, device=dev)
        gcnt = torch.zeros(ng * R, dtype=i64, device=dev)
        gcnt[pos] = counts
        gcnt2 = gcnt.view(ng, R)
        gofs = (torch.cumsum(gcnt2, 1) - gcnt2).reshape(-1)
        gtot = gcnt2.sum(1)
        fits = (usize <= CAP) & (gtot <= ecap) & (gcnt2.max(1).values <= 64)
        rec = torch.zeros((ng, words), dtype=torch.int32, device=dev)
        # column list, padded with the group's last column (an empty group reads row 0)
        last = torch.where(usize > 0, ucol[(ustart[1:] - 1).clamp(min=0)], torch.zeros_like(usize))
        rec[:, :CAP] = last.to(torch.int32).unsqueeze(1)
        uslot = torch.arange(uniq.numel(), device=dev) - ustart[ugrp]
        keep_u = fits[ugrp]
        rec.view(-1)[(ugrp[keep_u] * words + uslot[keep_u])] = ucol[keep_u].to(torch.int32)
        unfit_cols = ~fits
        if bool(unfit_cols.any()):
            rec[unfit_cols, :CAP] = 0
        # headers
        fit_row = fits.repeat_interleave(R)
        meta = torch.where(fit_row, gcnt | (gofs << 16), torch.full_like(gcnt, 0xffff))
        hdr = rec[:, CAP:E0].reshape(ng, R, 2)
        hdr[:, :, 0] = grow.view(ng, R).to(torch.int32)
        hdr[:, :, 1] = meta.view(ng, R).to(torch.int32)
        # entries of the staged groups
        keep_e = fits[e_grp]
        q = torch.arange(self.nnz, device=dev) - self.rowptr.to(i64)[rows]
        w = e_grp * words + E0 + 2 * (gofs[e_grp * R + e_i] + q)
        flat = rec.view(-1)
        flat[w[keep_e]] = slot[keep_e].to(torch.int32)
        flat[w[keep_e] + 1] = self.val[keep_e].view(torch.int32)
        staged_nnz = int(keep_e.sum()) if self.nnz else 0
        loads = float(usize[fits].sum() + (self.nnz - staged_nnz)) / max(n, 1)
        # the DMA waves read up to 2 D groups past the end of the walk of a workgroup only inside their own range, but
        # keep one spare record so that a plan for zero groups still has an address
        self.rec = {'rows': R, 'cap': CAP, 'kib': int(kib), 'groups': ng, 'rec': rec.contiguous(),
                    'loads_per_row': loads, 'staged': staged_nnz / max(self.nnz, 1)}
        self._view = None
        return self.rec['staged'], loads

    def detect_stencil_order(self, px=4, py=4, n_chunks=8):
        """If the operator is a 2-D lattice stencil in row-major node order (every entry's column is the row plus
        a * S + b with |a|, |b| <= 2 for one stride S - the reference's grid graphs, utils_in_learn_dynamics.py:137-157,
        handed over as plain tensors), return the group order that visits the lattice in px x py patches (the rows of a
        patch share most of their neighbours: 36 distinct columns for 16 rows of the 8-neighbour grid instead of 54):
        an int32 array of px * py slots per patch, -1 where a patch sticks out of the lattice.  None otherwise.  O(nnz)."""
        n = self.shape[0]
        # A row block of a sharded lattice (ndcn_amd/sharding.py) says where it sits: lattice_hint = (row_base, n_own) -
        # row r is node row_base + r of the shard, columns < n_own are the shard's own nodes (checked against the
        # stencil), columns >= n_own are halo rows (any: they only join the groups' column lists).
        row_base, n_own = getattr(self, 'lattice_hint', None) or (0, self.shape[1])
        if self.nnz == 0 or n < 64 or (getattr(self, 'lattice_hint', None) is None and self.shape[0] != self.shape[1]):
            return None
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(n, device=self.device), counts) + row_base
        cols = self.colidx.to(torch.int64)
        own = cols < n_own
        off = torch.unique((cols - rows)[own])
        if off.numel() > 25 or off.numel() == 0:
            return None
        off = off.cpu().numpy()
        big = np.abs(off)[np.abs(off) > 2]      # (a boundary band of a shard may see only the lattice row ABOVE it among its own columns)
        if big.size == 0:
            return None
        # the smallest large offset is S - b_max with b_max <= 2: of the three strides that allows take the one that
        # leaves the smallest in-row offsets
        S, best_b = 0, 3
        for cand in (int(big.min()), int(big.min()) + 1, int(big.min()) + 2):
            a = np.rint(off / cand)
            b = off - a * cand
            if cand >= 8 and np.all(np.abs(a) <= 2) and np.abs(b).max() < best_b:
                S, best_b = cand, int(np.abs(b).max())
        if S == 0:
            return None
        self.stencil_stride = S
        # patches in row-major patch order, each padded to px * py slots (-1) so that groups never straddle patches;
        # lattice coordinates are those of the shard (node = row_base + r), slots hold this operator's row indices
        x_lo, x_hi = row_base // S, (row_base + n - 1) // S + 1
        PX, PY = (x_hi - x_lo + px - 1) // px, (S + py - 1) // py
        x = x_lo + (np.arange(PX)[:, None, None, None] * px + np.arange(px)[None, None, :, None])
        y = (np.arange(PY)[None, :, None, None] * py + np.arange(py)[None, None, None, :])
        node = x * S + y
        node = np.where((x < x_hi) & (y < S) & (node >= row_base) & (node < row_base + n), node - row_base, -1)
        node = node.reshape(PX * PY, px * py)
        # Order of the patches: the group kernels (spmm_rec.hip, rhs_fused3.hip) give each of the 8 XCDs a contiguous
        # range of the walk and its 32 workgroups take the range round-robin - 32 consecutive patches run concurrently.
        # Row-major, those are 32 patches of one lattice row band and the band below comes PY patches later: its two
        # shared node rows have left the XCD's 4 MiB L2 by then (measured: X fetched 1.55 times per launch).  Inside
        # every XCD range the patches are therefore walked in strips `strip` patches wide, top to bottom: what runs
        # concurrently is one row of a strip, and the next iteration is the row right below it.
        strip = int(os.environ.get('NDCN_PATCH_STRIP', '32'))
        if strip > 0 and PX > 1 and PY > strip:
            t = np.arange(PX * PY, dtype=np.int64)
            X, Y = t // PY, t % PY
            per = (PX * PY + n_chunks - 1) // n_chunks
            node = node[np.lexsort((Y % strip, X, Y // strip, t // per))]
        return node.reshape(-1).astype(np.int32)

    def lattice_tile_order(self, S, block_rows=32, n_chunks=8):
        """Walk order of the fused RHS kernel's 64-row tiles on a lattice of stride S: the kernel gives each XCD a
        contiguous range of walk positions and its 32 workgroups take them round-robin, so 32 consecutive positions run
        concurrently - here they form a block of `block_rows` lattice rows x 64 columns, whose neighbour rows
        (34 x 66 nodes, 2.2 MB at H = 256) fit the XCD's 4 MiB L2: every X row is then fetched from HBM about once per
        launch instead of 1.3-1.6 times.  Within an XCD's range the blocks follow each other along the lattice row band."""
        n = self.shape[0]
        nt = (n + 63) // 64
        t = np.arange(nt, dtype=np.int64)
        x, y = (64 * t) // S, (64 * t) % S                  # lattice coordinates of a tile's first node
        per = (nt + n_chunks - 1) // n_chunks
        band = t // per                                     # keep the XCD ranges where the kernel cuts them
        key = ((band * (n // S // block_rows + 2) + x // block_rows) * (S // 64 + 2) + y // 64) * block_rows + x % block_rows
        return np.argsort(key, kind='stable').astype(np.int32)

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
        }
        # The hubs' rows: scratch the fused kernel reads as (part of) its second panel.  For a shard with a halo panel
        # (n_halo > 0: HaloPlan sets it before the plan is built) ONE buffer holds [halo rows | hub rows]: the exchange
        # receives into its head, 'S' is its tail - the layout the C side accepts next to a halo panel (ndcn_hip.h).
        n_halo = int(getattr(self, 'n_halo', 0))
        self.hub['halo_S'] = torch.empty(n_halo + hubs.size, H, dtype=torch.float32, device=dev)
        self.hub['S'] = self.hub['halo_S'][n_halo:]
        return int(hubs.size)

    def ensure_plans(self, H):
        """One-off, lazy: attach the operator plans the H = 256 kernels use - the long-row plan for skewed degree
        distributions and the group-record plan when neighbouring rows share enough neighbours for staging to pay."""
        if H != 256 or getattr(self, '_plans_tried', False):
            return self
        self._plans_tried = True
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
        if os.environ.get('NDCN_REC_PLAN', '1') == '0' or self.nnz == 0:
            return self
        # Group-record plan: with a lattice walk order (given by the caller or detected) 16-row patches, otherwise 8
        # consecutive rows; kept when it covers the operator and stages clearly fewer rows than a direct gather fetches.
        if self.group_order is None and self.row_order is None and os.environ.get('NDCN_REC_STENCIL', '1') != '0':
            order = self.detect_stencil_order()
            if order is not None:
                self.group_order = torch.as_tensor(order, dtype=torch.int32).to(self.device)
                if os.environ.get('NDCN_TILE_ORDER', '1') != '0' and getattr(self, 'lattice_hint', None) is None:
                    self.tile_order = torch.as_tensor(self.lattice_tile_order(self.stencil_stride), dtype=torch.int32).to(self.device)
        avg = self.nnz / max(self.shape[0], 1)
        best = None
        hinted = self.group_order is not None or self.row_order is not None
        # with a walk order: the lattice shapes; without: 8 consecutive rows with a 32-column list, or - when that covers too
        # few groups (ring neighbours + random shortcuts: a small world's 8 rows reference ~28 distinct columns) - 48
        for shape in (self.REC_SHAPES[1::-1] if hinted else (self.REC_SHAPES[0], self.REC_SHAPES[2])):
            staged, loads = self.build_rec_plan(*shape)
            if staged >= 0.9 and loads <= 0.75 * avg and (best is None or loads < best[1]):
                best = (self.rec, loads)
        self.rec = best[0] if best is not None else None
        self._view = None
        return self



def sweep_plan_reference(rowptr, colidx, val, shape):
    """Column-sweep plan (include/ndcn_hip.h, struct ndcn_csr: sweep_*; csrc/csr_plan.hip: build_sweep_plan) restated in numpy:
    passes of <= 100 352 rows, 8 XCD chunks per pass, 256 slabs of consecutive rows per chunk; per slab the entries merged over
    its rows, stably sorted by column, packed as {row in slab << 24 | column, value bits} and padded to groups of 8 with
    {49 << 24, 0}.  Returns dict(passes, rows_per_pass, rpw, slab [n_slabs, 2] int32, ent [n_entries + 32, 2] uint32)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx, dtype=np.int64)
    bits = np.asarray(val, dtype=np.float32).view(np.uint32)
    n = int(shape[0])
    cap = 8 * 256 * 49
    passes = (n + cap - 1) // cap
    rpp = (n + passes - 1) // passes
    slab, ent = [], []
    pos = 0
    for p in range(passes):
        base, end = p * rpp, min(n, (p + 1) * rpp)
        per_xcd = (end - base + 7) // 8
        rpw = (per_xcd + 255) // 256
        for x in range(8):
            xend = min(end, base + (x + 1) * per_xcd)
            for sl in range(256):
                r0 = min(xend, base + x * per_xcd + sl * rpw)
                r1 = min(xend, r0 + rpw)
                lo, hi = rowptr[r0], rowptr[r1]
                rows = np.repeat(np.arange(r0, r1), np.diff(rowptr[r0:r1 + 1])) - r0
                order = np.argsort(colidx[lo:hi], kind='stable')
                key = (rows[order].astype(np.uint32) << np.uint32(24)) | colidx[lo:hi][order].astype(np.uint32)
                cnt = int(hi - lo)
                pad = (-cnt) % 8
                ent.append(np.stack([key, bits[lo:hi][order]], 1))
                if pad:
                    ent.append(np.tile(np.array([[49 << 24, 0]], dtype=np.uint32), (pad, 1)))
                slab.append((pos, cnt))
                pos += cnt + pad
    ent.append(np.tile(np.array([[49 << 24, 0]], dtype=np.uint32), (32, 1)))
    per_xcd0 = (rpp + 7) // 8
    return {'passes': passes, 'rows_per_pass': rpp, 'rpw': (per_xcd0 + 255) // 256, 'entries': pos,
            'slab': np.asarray(slab, dtype=np.int32).reshape(-1, 2), 'ent': np.concatenate(ent).astype(np.uint32)}
