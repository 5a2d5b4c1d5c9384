"""Node-range sharding of the hot path across the GPUs of one node.

Rank r owns a contiguous range of nodes: the matching rows of the operator and of every state panel.
Everything in a solver step is row-local except A X, which needs the X rows of remote column neighbours
("halo").  Per RHS evaluation: pack the rows other ranks need (HIP gather kernel), ONE all-to-all-v over
RCCL (point-to-point sends over xGMI; only the referenced rows travel, not an all-gather of the panel),
then the local SpMM reads own rows and halo rows from two panels (ndcn_spmm_f32's X / X_halo).  The
adaptive controller needs two global scalars per step (sum of squared error ratios, non-finite count):
one 16-byte all-reduce.  All ranks see identical scalars, hence take identical accept/reject decisions.

torch.distributed is the transport ("nccl" = RCCL on ROCm; "gloo" in the CPU tests, which drive this same
code with the oracle-backed ops double).
"""
import os

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.nn as nn

from .csr import CsrOperator


def even_bounds(n, world):
    """Contiguous node ranges: [bounds[r], bounds[r+1])."""
    return [(n * r) // world for r in range(world + 1)]


class HaloPlan:
    """What this rank must send / receive before each A X, and the operator remapped to [own | halo] columns."""

    def __init__(self, rows_block, bounds, rank, device, group=None, self_halo=0, two_phase=None):
        """rows_block: scipy CSR, this rank's rows x ALL global columns.
        self_halo (test hook, NDCN_SELF_HALO): the first `self_halo` OWN columns are additionally routed through the
        exchange as if a peer owned them (this rank sends them to itself), so that a single rank drives a non-empty
        all-to-all-v over the real backend - on a 1-GPU box that is the only way RCCL's collective ever executes."""
        self.group = group
        self.rank, self.world = rank, len(bounds) - 1
        if two_phase is None:
            two_phase = os.environ.get('NDCN_TWO_PHASE', '1') != '0'
        self.bounds = list(bounds)
        lo, hi = bounds[rank], bounds[rank + 1]
        self.n_own = hi - lo
        blk = rows_block.tocsr()
        blk.sort_indices()
        assert blk.shape[0] == self.n_own
        cols = blk.indices.astype(np.int64)
        remote = (cols < lo) | (cols >= hi) | self._self_halo_mask(cols - lo, self_halo)
        need = np.unique(cols[remote])                                   # sorted global ids = halo order
        owner = np.searchsorted(np.asarray(bounds[1:]), need, side='right')
        self.recv_counts = [int((owner == p).sum()) for p in range(self.world)]
        self.n_halo = int(need.size)
        # remap columns: own -> c - lo ; remote -> n_own + position in `need`
        new_cols = np.where(remote, self.n_own + np.searchsorted(need, cols), cols - lo)
        local = sp.csr_matrix((blk.data.astype(np.float32), new_cols, blk.indptr),
                              shape=(self.n_own, self.n_own + self.n_halo))
        local.sort_indices()
        self.local_op = CsrOperator.from_scipy(local, device)
        self.local_op.n_halo = self.n_halo                               # the long-row plan lays its scratch out behind the halo rows
        self.local_op.lattice_hint = (0, self.n_own)                     # columns >= n_own are halo rows
        self.local_nnz = int(local.nnz)
        # Row ranges for overlapping the exchange with compute: rows that reference no halo column ("interior") can be
        # evaluated while the halo is in flight.  With node-range sharding of a graph in a locality-preserving order they
        # form one long run between two thin boundary bands (grid: all but the first and last lattice row of the shard).
        self.ranges = self._row_ranges(local, device)
        # Shards whose halo columns are scattered over all rows (small-world shortcuts, power-law graphs) have no interior
        # run to hide the exchange behind.  They evaluate A X in two phases instead, A = [A_own | A_halo]:
        #   phase 1 (while the all-to-all-v is in flight)  S = A_own X                          own columns only
        #   phase 2 (halo landed)                          the fused RHS on [I | A_halo] over the panels [S | X_halo]
        # Row i of the phase-2 operator is {(i, 1.0)} followed by the row's halo entries: the fold starts from S_i and adds
        # the halo products in the stored order - the very sequence of fma's the one-launch form performs (halo columns
        # sort behind the own ones), so the two forms agree bit for bit (up to the sign of a zero).
        self.two_phase = self._two_phase_ops(local, device) if two_phase and self.ranges is None and self.n_halo > 0 else None
        # tell every owner which of its rows we need (plan-time exchange of index lists)
        want = [need[owner == p] - bounds[p] for p in range(self.world)]     # owner-local row ids
        self.send_counts, send_idx = self._exchange_requests(want, device)
        self.send_idx = send_idx.to(torch.int32)                             # rows of OUR panel to pack, grouped by peer
        self.device = device
        # Whether ANY rank moves a row: a collective may only be skipped on a fact every rank agrees on (a rank whose
        # shard happens to have no cross-shard edge - a disconnected component, a block-diagonal graph - must still
        # enter the all-to-all its peers enter, with zero counts).
        tot = torch.tensor([self.n_halo + sum(self.send_counts)], dtype=torch.int64,
                           device=device if dist.get_backend(group) == 'nccl' else torch.device('cpu'))
        if self.world > 1:
            dist.all_reduce(tot, group=group)
        self.global_rows_moved = int(tot.item())

    def _self_halo_mask(self, own_cols, spec):
        """Own columns routed through the exchange by the self-halo hook: an int k = the first k own columns (a lattice
        shard's boundary band), 'scatter:k' = k own columns drawn uniformly (seed 0) - a halo as scattered as a
        small-world shard's, so that a single GPU exercises the two-phase evaluation against a real RCCL exchange."""
        if isinstance(spec, str) and spec.startswith('scatter:'):
            k = min(int(spec.split(':')[1]), self.n_own)
            pick = np.zeros(self.n_own + 1, dtype=bool)
            pick[np.random.RandomState(0).choice(self.n_own, size=k, replace=False)] = True
            inside = (own_cols >= 0) & (own_cols < self.n_own)
            return inside & pick[np.clip(own_cols, 0, self.n_own)]
        return (own_cols >= 0) & (own_cols < int(spec or 0))

    def _two_phase_ops(self, local, device):
        n = self.n_own
        own = local[:, :n].tocsr()
        own_op = CsrOperator.from_scipy(own, device)
        halo_part = local[:, n:].tocsr()
        eye = sp.identity(n, dtype=np.float32, format='csr')
        second = sp.hstack([eye, halo_part], format='csr')
        second.sort_indices()
        halo_op = CsrOperator.from_scipy(second, device)
        halo_op.n_halo = self.n_halo                        # long-row plan: scratch behind the halo rows (as local_op)
        return own_op, halo_op

    def _row_ranges(self, local, device):
        """[(a, b, operator of rows [a, b), needs_halo)] covering the shard, or None when no long interior run exists."""
        n = self.n_own
        if n == 0 or self.n_halo == 0:
            return None
        has_halo = np.zeros(n, dtype=bool)
        rows = np.repeat(np.arange(n), np.diff(local.indptr))
        has_halo[rows[local.indices >= n]] = True
        idx = np.flatnonzero(has_halo)
        if idx.size == 0:
            return None
        # longest run of interior rows
        edges = np.concatenate([[-1], idx, [n]])
        gaps = np.diff(edges) - 1
        g = int(np.argmax(gaps))
        a, b = int(edges[g] + 1), int(edges[g + 1])
        if b - a < 0.5 * n:
            return None
        out = []
        for lo_, hi_, halo in ((0, a, True), (a, b, False), (b, n, True)):
            if hi_ <= lo_:
                continue
            sub = local[lo_:hi_]
            if not halo:
                sub = sub[:, :n]                                    # interior rows reference own columns only
            op = CsrOperator.from_scipy(sub, device)
            op.lattice_hint = (lo_, n)                              # where the row block sits in the shard (ndcn_csr_hints::lattice_row_base / lattice_n_own)
            out.append((lo_, hi_, op, halo))
        return out

    def _exchange_requests(self, want, device):
        world = self.world
        comm_dev = device if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        counts_out = torch.tensor([len(w) for w in want], dtype=torch.int64, device=comm_dev)
        counts_in = torch.empty(world, dtype=torch.int64, device=comm_dev)
        dist.all_to_all_single(counts_in, counts_out, group=self.group)
        send_counts = [int(c) for c in counts_in.tolist()]
        flat_out = torch.from_numpy(np.concatenate(want).astype(np.int64) if world else np.empty(0, np.int64)).to(comm_dev)
        flat_in = torch.empty(sum(send_counts), dtype=torch.int64, device=comm_dev)
        dist.all_to_all_single(flat_in, flat_out, send_counts, [int(c) for c in counts_out.tolist()], group=self.group)
        return send_counts, flat_in.to(device)

    def bytes_per_exchange(self, H):
        return 4 * H * (sum(self.send_counts) + self.n_halo)

    def exchange(self, ops, X):
        """Returns the halo panel (n_halo x H) for the local panel X."""
        H = X.shape[1]
        hub = getattr(self.two_phase[1] if self.two_phase is not None else self.local_op, 'hub', None)
        if hub is not None and hub['H'] == H and hub['halo_S'].shape[0] == self.n_halo + hub['n'] and X.is_cuda:
            # long-row plan on this shard: the halo rows land in the head of the plan's [halo | hub rows] buffer (the
            # next exchange cannot start before every launch that reads it has produced its part of the next panel)
            halo = hub['halo_S'][:self.n_halo]
        else:
            halo = torch.empty((self.n_halo, H), dtype=X.dtype, device=X.device)
        if self.global_rows_moved == 0:                                   # nobody sends or receives: no collective on ANY rank
            return halo
        packed = ops.gather_rows(X, self.send_idx) if self.send_idx.numel() else X[:0]
        if X.is_cuda and dist.get_backend(self.group) != 'nccl':
            # gloo moves host memory only: stage through the host (the 2-rank GPU test runs both ranks on one device,
            # which RCCL refuses; production is nccl = RCCL, device to device over xGMI)
            h = torch.empty((self.n_halo, H), dtype=X.dtype)
            dist.all_to_all_single(h, packed.cpu(), self.recv_counts, self.send_counts, group=self.group)
            halo.copy_(h)
            return halo
        dist.all_to_all_single(halo, packed, self.recv_counts, self.send_counts, group=self.group)
        return halo


    # ---- training: the exchange's vector-Jacobian product (round 6) ---------------------------------------------------------------
    def _scatter_operator(self):
        """P^T: (n_own x rows we send) with a one where packed row j is own row send_idx[j] - the accumulation of the returning
        gradient rows as an SpMM: a fixed order per own row (an index_add_ with atomics would not be deterministic)"""
        if getattr(self, '_scatter_op', None) is None:
            idx = self.send_idx.cpu().numpy().astype(np.int64)
            m = sp.csr_matrix((np.ones(idx.size, dtype=np.float32), (idx, np.arange(idx.size))), shape=(self.n_own, max(idx.size, 1)))
            m.sort_indices()
            self._scatter_op = CsrOperator.from_scipy(m, self.device)
        return self._scatter_op

    def exchange_grad(self, ops, G):
        """The backward of `exchange`: G (n_halo x H) is the gradient of the halo panel this rank received; every row goes back to its
        owner (the same all-to-all-v with the counts swapped) and the owner adds what returns into the rows it had sent.  Returns this
        rank's (n_own x H) share: the gradient that reaches its own panel THROUGH the other ranks' halos.  Collective: every rank of
        the group calls it (autograd does: the ranks run the same graph in the same order)."""
        H = G.shape[1]
        out_rows = sum(self.send_counts)
        if self.global_rows_moved == 0:
            return torch.zeros((self.n_own, H), dtype=G.dtype, device=G.device)
        back = torch.empty((out_rows, H), dtype=G.dtype, device=G.device)
        G = G.contiguous()
        if G.is_cuda and dist.get_backend(self.group) != 'nccl':
            h = torch.empty((out_rows, H), dtype=G.dtype)
            dist.all_to_all_single(h, G.cpu(), self.send_counts, self.recv_counts, group=self.group)
            back.copy_(h)
        else:
            dist.all_to_all_single(back, G, self.send_counts, self.recv_counts, group=self.group)
        if out_rows == 0:
            return torch.zeros((self.n_own, H), dtype=G.dtype, device=G.device)
        return ops.spmm(self._scatter_operator(), back)


class _HaloExchangeFn(torch.autograd.Function):
    """halo = exchange(x) as a node of the training graph: backward = the reverse all-to-all-v with accumulation (HaloPlan.exchange_grad)"""

    @staticmethod
    def forward(ctx, x, plan, ops):
        ctx.plan, ctx.ops = plan, ops
        return plan.exchange(ops, x.detach()).clone()        # (the plan may hand out a view of a buffer it re-uses: the graph keeps its own)

    @staticmethod
    def backward(ctx, g):
        return ctx.plan.exchange_grad(ctx.ops, g), None, None


def allreduce_gradients(params, group=None):
    """Parameters are replicated, every rank's backward leaves the contribution of ITS rows in `.grad`: the sum over ranks is the
    gradient of a loss summed over all nodes (call once after loss.backward(); heat_dynamics.py:313-334 on a sharded graph)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        g = p.grad
        if g.is_cuda and dist.get_backend(group) != 'nccl':
            h = g.cpu()
            dist.all_reduce(h, group=group)
            g.copy_(h)
        else:
            dist.all_reduce(g, group=group)


class ShardedODEFunc(nn.Module):
    """ODEFunc on one shard: halo exchange, then the local fused RHS over [own | halo]
    (neural_dynamics.py:20-39 semantics on the global graph).

    Overlap of the exchange (side stream) with compute (caller's stream), two forms:
      * row split - the shard has a long run of interior rows (HaloPlan.ranges; lattices): the interior rows - a launch
        that needs no halo panel - run during the exchange, the two thin boundary bands follow once the halo has landed.
        Every launch mode is split, the dopri5 error record too: its launches take the rows of y1 they own explicitly
        (`y1=`) and add their {sum, bad} into one device record (`accum=`), read back once.
      * two-phase - the halo columns are scattered over all rows (HaloPlan.two_phase; small-world, power-law shards):
        S = A_own X runs during the exchange, then ONE fused launch on [I | A_halo] over [S | X_halo] finishes A X and
        carries the Linear and the RK epilogue.
    Results are identical to the un-split evaluation of the shard (rows are independent; the two-phase fold performs the
    same fma sequence)."""

    ndcn_autonomous = True
    supports_aux = True               # rhs_rk(..., aux_cs=) forms dopri5's partial error sum in the stage-6 launch

    def __init__(self, odefunc, plan, ops, overlap=True):
        super().__init__()
        self.f = odefunc
        self.plan = plan
        self.ops = ops
        self.nfe = 0
        self.halo_bytes = 0
        self.overlap = overlap and plan.ranges is not None
        self.two_phase = overlap and plan.two_phase is not None
        self.comm = None                  # side stream of the exchange (created on first use, CUDA only)
        self._err_record = None           # this function's own error record (split launches accumulate into it)
        self.timing = None                # when a dict: accumulates {exchange_us, interior_us, exposed_us, n}

    # -- exchange on the side stream; returns (halo, event-or-None)
    def _start_exchange(self, x):
        self.halo_bytes += self.plan.bytes_per_exchange(x.shape[1])
        if not ((self.overlap or self.two_phase) and x.is_cuda):
            return self.plan.exchange(self.ops, x), None
        import torch.cuda as tc
        if self.comm is None:
            self.comm = tc.Stream(device=x.device)
        cur = tc.current_stream(x.device)
        self.comm.wait_stream(cur)                                   # x is complete
        ev = None
        with tc.stream(self.comm):
            if self.timing is not None:
                e0 = tc.Event(enable_timing=True)
                e0.record()
            halo = self.plan.exchange(self.ops, x)
            halo.record_stream(cur)
            x.record_stream(self.comm)
            if self.timing is not None:
                e1 = tc.Event(enable_timing=True)
                e1.record()
                ev = (e0, e1)
        return halo, ev

    def _finish_exchange(self, x, ev, t_int):
        if self.comm is None or not x.is_cuda:
            return
        import torch.cuda as tc
        cur = tc.current_stream(x.device)
        if self.timing is not None and ev is not None:
            i0, i1 = t_int
            i1.record()
        cur.wait_stream(self.comm)
        if self.timing is not None and ev is not None:
            w = tc.Event(enable_timing=True)
            w.record()
            self.timing.setdefault('pending', []).append((ev[0], ev[1], i0, i1, w))

    def drain_timing(self):
        """Fold the recorded events into microsecond sums (call after a device synchronise)."""
        t = self.timing
        if not t:
            return t
        for e0, e1, i0, i1, w in t.pop('pending', []):
            t['exchange_us'] = t.get('exchange_us', 0.0) + 1e3 * e0.elapsed_time(e1)
            t['interior_us'] = t.get('interior_us', 0.0) + 1e3 * i0.elapsed_time(i1)
            t['exposed_us'] = t.get('exposed_us', 0.0) + 1e3 * max(0.0, i1.elapsed_time(w))
            t['n'] = t.get('n', 0) + 1
        return t

    def _overlapped(self, x, during, after):
        """during() runs on the caller's stream while the exchange of x is in flight; after(halo) once it has landed."""
        halo, ev = self._start_exchange(x)
        t_int = None
        if self.timing is not None and ev is not None:
            import torch.cuda as tc
            t_int = (tc.Event(enable_timing=True), tc.Event(enable_timing=True))
            t_int[0].record()
        mid = during()
        self._finish_exchange(x, ev, t_int)
        return after(halo, mid)

    def _split_eval(self, x, call):
        """call(op, X_halo, a, b, first, last) evaluates rows [a, b) of the shard; interior first, boundary after the
        exchange."""
        rr = self.plan.ranges
        inner = [r for r in rr if not r[3]]
        outer = [r for r in rr if r[3]]

        def during():
            for i, (a, b, op, _) in enumerate(inner):
                call(op, None, a, b, i == 0, False)

        def after(halo, _):
            out = None
            for i, (a, b, op, _) in enumerate(outer):
                out = call(op, halo, a, b, not inner and i == 0, i == len(outer) - 1)
            return out
        return self._overlapped(x, during, after)

    def _training(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.f.parameters()))

    def _forward_with_grad(self, x):
        """The evaluation as nodes of an autograd graph (round 6: training on a sharded graph): the exchange is differentiable
        (_HaloExchangeFn: its backward sends the halo rows' gradients home), the local right-hand side runs on [own | halo] as one
        panel through the differentiable op set - A^T g in its backward is the local operator's transpose, (n_own + n_halo) rows.
        One launch form, no overlap: correctness first (the reference trains by plain backpropagation, heat_dynamics.py:313-334)."""
        f = self.f
        W, bias = f.wt.weight, f.wt.bias
        ops = self.ops
        native = getattr(ops, 'differentiable', False)       # (the CPU test double is plain torch: differentiable as it stands)
        if f.no_graph:
            if native:
                return ops.rhs(None, x, W, bias, no_graph=True, no_control=f.no_control)
            from .autograd_ops import rhs as rhs_g
            return rhs_g(None, x, W, bias, True, f.no_control)
        halo = _HaloExchangeFn.apply(x, self.plan, ops)
        self.halo_bytes += 2 * self.plan.bytes_per_exchange(x.shape[1])          # (there and, in the backward, back)
        if native:
            return ops.rhs(self.plan.local_op, x, W, bias, no_control=f.no_control, X_halo=halo)
        from .autograd_ops import rhs as rhs_g
        return rhs_g(self.plan.local_op, torch.cat([x, halo], 0), W, bias, False, f.no_control)

    def forward(self, t, x):
        self.nfe += 1
        if self._training(x):
            return self._forward_with_grad(x)
        f = self.f
        W, bias = f.wt.weight, f.wt.bias
        if f.no_graph:
            return self.ops.rhs(None, x, W, bias, no_graph=True, no_control=f.no_control)
        if self.overlap:
            out = torch.empty_like(x)
            self._split_eval(x, lambda op, halo, a, b, first, last: self.ops.rhs(op, x, W, bias, no_control=f.no_control,
                                                                                  X_halo=halo, out=out[a:b]))
            return out
        if self.two_phase:
            own_op, halo_op = self.plan.two_phase
            return self._overlapped(x, lambda: self.ops.spmm(own_op, x),
                                    lambda halo, S: self.ops.rhs(halo_op, S, W, bias, no_control=f.no_control, X_halo=halo))
        halo, _ = self._start_exchange(x)
        return self.ops.rhs(self.plan.local_op, x, W, bias, no_control=f.no_control, X_halo=halo)

    def rhs_rk(self, x, mode, y0, kprev, cs, rtol, atol, aux_cs=None):
        """RHS + the stage algebra consuming it (core.Dopri5's `fused` protocol); the error record is summed
        over ranks here so every rank sees the same controller input.  aux_cs (combine): also the second linear
        combination sum aux_cs[m] kprev[m] + aux_cs[-1] K -> returns (k, y_next, aux)."""
        self.nfe += 1
        f = self.f
        W, bias = f.wt.weight, f.wt.bias
        akw = lambda a=None, b=None: {} if aux_cs is None else {'aux_cs': aux_cs, 'out_aux': aux if a is None else aux[a:b]}
        aux = torch.empty_like(x) if aux_cs is not None else None
        if self.overlap and not f.no_graph:
            k = torch.empty_like(x)
            y_next = torch.empty_like(x) if mode in ('combine', 'rk4') else None

            def call(op, halo, a, b, first, last):
                if mode == 'error':
                    if self._err_record is None:
                        self._err_record = self.ops.new_error_record(x.device)
                    return self.ops.rhs_rk(op, x, W, bias, mode, y0[a:b], [kp[a:b] for kp in kprev], cs, rtol, atol,
                                           no_control=f.no_control, X_halo=halo, out_K=k[a:b], y1=x[a:b], accum=not first,
                                           fetch=last, record=self._err_record)[1]
                self.ops.rhs_rk(op, x, W, bias, mode, y0[a:b], [kp[a:b] for kp in kprev], cs, rtol, atol,
                                no_control=f.no_control, X_halo=halo, out_K=k[a:b], out_y=y_next[a:b], **akw(a, b))
            out = self._split_eval(x, call)
            if mode != 'error':
                return (k, y_next) if aux is None else (k, y_next, aux)
        elif self.two_phase and not f.no_graph:
            own_op, halo_op = self.plan.two_phase
            res = self._overlapped(x, lambda: self.ops.spmm(own_op, x),
                                   lambda halo, S: self.ops.rhs_rk(halo_op, S, W, bias, mode, y0, kprev, cs, rtol, atol,
                                                                   no_control=f.no_control, X_halo=halo,
                                                                   y1=x if mode == 'error' else None, **akw()))
            if mode != 'error':
                return res
            k, out = res
        else:
            halo = None
            if not f.no_graph:
                halo, _ = self._start_exchange(x)
            res = self.ops.rhs_rk(None if f.no_graph else self.plan.local_op, x, W, bias, mode, y0, kprev, cs,
                                  rtol, atol, no_graph=f.no_graph, no_control=f.no_control, X_halo=halo, **akw())
            if mode != 'error':
                return res
            k, out = res
        if mode == 'error' and self.plan.world > 1:
            dev = x.device if dist.get_backend(self.plan.group) == 'nccl' else torch.device('cpu')
            v = torch.tensor([out[0], out[1]], dtype=torch.float64, device=dev)
            dist.all_reduce(v, group=self.plan.group)
            out = tuple(v.tolist())
        return k, out


class DistOps:
    """Panel ops whose reductions span all ranks (everything else is row-local and forwarded untouched)."""

    def __init__(self, base, n_global_rows, n_local_rows, group=None):
        self.base = base
        self.group = group
        self.ratio = n_global_rows / float(n_local_rows) if n_local_rows else 1.0
        self.n_global_rows, self.n_local_rows = n_global_rows, n_local_rows
        self.name = 'dist(%s)' % getattr(base, 'name', '?')

    def __getattr__(self, item):
        return getattr(self.base, item)

    def numel(self, t):
        return (t.numel() // self.n_local_rows) * self.n_global_rows if self.n_local_rows else 0

    def _allreduce(self, s, bad, like):
        dev = like.device if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        v = torch.tensor([s, bad], dtype=torch.float64, device=dev)
        dist.all_reduce(v, group=self.group)
        s, bad = v.tolist()
        return s, bad

    def error(self, y0, y1, ks, cs, rtol, atol):
        s, bad = self.base.error(y0, y1, ks, cs, rtol, atol)
        return self._allreduce(s, bad, y0)

    def scaled_sumsq(self, a, b, y, rtol, atol):
        s, bad = self.base.scaled_sumsq(a, b, y, rtol, atol)
        return self._allreduce(s, bad, a)


def sharded_odeint(ops, odefunc, plan, n_global_rows, x_local, t, rtol=1e-7, atol=1e-9, method='dopri5', step_log=None,
                   group=None, fused=True, stats=None):
    """odeint on this rank's rows of the global system; returns (len(t), n_local, H).
    stats (a dict, optional) receives which evaluation form ran and what travelled: {'form': 'row_split' | 'two_phase' |
    'one_launch', 'nfe', 'halo_bytes' (sent + received over the solve), 'halo_rows_received_per_rhs'}."""
    from .torchdiffeq._impl import core
    f = ShardedODEFunc(odefunc, plan, ops)
    training = f._training(x_local)
    if training:
        # Training (round 6): every panel operation of the solve is a node of this rank's autograd graph - the differentiable op set
        # (autograd_ops on the device; the CPU double is torch as it stands), the halo exchange with its reverse exchange as backward.
        # Fixed grids: the reference's gradient.  dopri5: the step sizes the (globally reduced) controller chose are constants of the
        # graph - unlike the single-GPU tape, which follows the reference through the controller; stated in DESIGN section 6.
        # Parameter gradients are per-rank contributions: allreduce_gradients() after backward.
        if not getattr(ops, 'differentiable', False):
            from .autograd_ops import autograd_ops
            ops = autograd_ops
        fused = False
    dops = DistOps(ops, n_global_rows, x_local.shape[0], group)
    _, func, y0, tt = core.check_inputs(f, x_local, t)
    # A decreasing grid is integrated as -f(-t, y) on -t (misc.py:184-189): the sign flip cannot ride in the ReLU
    # epilogue of the fused RHS, so those solves step through the un-fused stage kernels.
    decreasing = bool((t[1:] < t[:-1]).all())
    if method == 'dopri5':
        sol = core.integrate_dopri5(dops, func, y0, tt, rtol, atol, autonomous=True, step_log=step_log,
                                    fused=f if (fused and not decreasing) else None)
    else:
        sol = core.integrate_fixed(dops, func, y0, tt, method, autonomous=True)
    if stats is not None:
        stats.update(form='one_launch+autograd' if training else 'row_split' if f.overlap else 'two_phase' if f.two_phase else 'one_launch', nfe=f.nfe,
                     halo_bytes=f.halo_bytes, halo_rows_received_per_rhs=plan.n_halo)
    return torch.stack([s[0] for s in sol])


class DeviceShard:
    """The C-ABI form of a HaloPlan (include/ndcn_hip.h: ndcn_comm / ndcn_halo_plan / ndcn_shard): an RCCL communicator
    of the library's own (unique id from rank 0, handed out over torch.distributed), the halo plan, and the way the
    shard is evaluated - row split for lattices, two-phase for scattered halos, one launch otherwise.  With it
    `DeviceSolver(..., shard=)` steps the whole solve inside libndcn_hip.so: exchange on a side stream, launches, the
    16-byte all-reduce and the controller - no Python between the evaluations."""

    def __init__(self, plan, n_global_rows, group=None, transport=None):
        import ctypes
        import time
        from . import _lib
        self.plan = plan
        self.lib = lib = _lib.load()
        self._pid = os.getpid()
        self.n_global_rows = int(n_global_rows)
        dev = plan.device
        # communicator: rank 0 draws the id, everybody learns it over the caller's process group.  Every rank reaches the
        # broadcast whatever happened before it (a rank that raised earlier would leave its peers waiting in it): rank 0's
        # failure travels as an all-zero id with a set flag byte, and then every rank raises together.
        self.comm = ctypes.c_void_p()
        self.transport = transport or os.environ.get('NDCN_COMM_TRANSPORT', 'rccl')
        if self.transport == 'loopback':
            # TEST transport (include/ndcn_hip.h: ndcn_comm_create_loopback): host-staged shared memory with the RCCL communicator's
            # call sequence - several ranks on ONE device.  Rank 0 draws the name, everybody learns it over the caller's group.
            name = [('ndcn_lb_%d_%x' % (os.getpid(), int(time.time() * 1e6) & 0xffffffffff)) if plan.rank == 0 else None]
            if plan.world > 1:
                dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            with torch.cuda.device(dev):
                _lib.check(lib.ndcn_comm_create_loopback(name[0].encode(), plan.world, plan.rank, ctypes.byref(self.comm)))
        else:
            idbuf = ctypes.create_string_buffer(128)
            rc0 = lib.ndcn_comm_unique_id(idbuf) if plan.rank == 0 else 0
            if plan.world > 1:
                comm_dev = dev if dist.get_backend(group) == 'nccl' else torch.device('cpu')
                t = torch.tensor(list(idbuf.raw) + [1 if rc0 != 0 else 0], dtype=torch.uint8, device=comm_dev)
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                raw = bytes(t.cpu().tolist())
                if raw[128]:
                    raise _lib.NdcnHipError(_lib.EHIP, 'rank 0 could not draw an RCCL unique id')
                idbuf = ctypes.create_string_buffer(raw[:128], 128)
            else:
                _lib.check(rc0)
            with torch.cuda.device(dev):
                _lib.check(lib.ndcn_comm_create(idbuf, plan.world, plan.rank, ctypes.byref(self.comm)))
        L = ctypes.c_int64 * plan.world
        self.send_idx = plan.send_idx.contiguous()
        self.halo = ctypes.c_void_p()
        _lib.check(lib.ndcn_halo_plan_create(self.comm, plan.n_halo, L(*plan.send_counts), L(*plan.recv_counts),
                                             _lib.ptr(self.send_idx) if self.send_idx.numel() else None,
                                             1 if plan.global_rows_moved > 0 else 0, ctypes.byref(self.halo)))
        self._views = {}

    def operator(self, H):
        """The operator the solver descriptor carries: [I | A_halo] in the two-phase form, the whole shard otherwise."""
        op = self.plan.two_phase[1] if self.plan.two_phase is not None else self.plan.local_op
        return op.ensure_plans(H)

    def view_ptr(self, H):
        import ctypes
        from . import _lib
        v = self._views.get(H)
        if v is None:
            p = self.plan
            v = _lib.ShardView()
            v.comm, v.halo, v.n_global_rows = self.comm, self.halo, self.n_global_rows
            keep = []
            if p.ranges is not None:
                v.n_blocks = len(p.ranges)
                for i, (a, b, op, needs) in enumerate(p.ranges):
                    op.ensure_plans(H)
                    v.blocks[i].row_lo, v.blocks[i].row_hi, v.blocks[i].needs_halo = a, b, 1 if needs else 0
                    v.blocks[i].A = op.view()
                    keep.append(op)
            elif p.two_phase is not None:
                own = p.two_phase[0].ensure_plans(H)
                v.A_own = own.view()
                keep.append(own)
            main = self.operator(H)
            hub = getattr(main, 'hub', None)
            if hub is not None and hub['H'] == H and hub['halo_S'].shape[0] == p.n_halo + hub['n']:
                v.X_halo = hub['halo_S'].data_ptr()             # long-row plan: the halo lands in front of its hub rows
            v._keep = keep
            self._views[H] = v
        return ctypes.pointer(v)

    def close(self):
        mine = getattr(self, '_pid', None) == os.getpid()     # (never from a fork()ed copy of this object)
        if self.halo and mine:
            self.lib.ndcn_halo_plan_destroy(self.halo)
        self.halo = None
        if self.comm and mine:
            self.lib.ndcn_comm_destroy(self.comm)
        self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedDeviceBench:
    """bench.py's N > 1 workload on the device-resident sharded solver (the production form over RCCL): the counterpart of
    ShardedBench without Python between the evaluations."""

    def __init__(self, odefunc, block, bounds, rank, device, T, rtol, atol, group=None, plan=None):
        from .torchdiffeq._impl.odeint import DeviceSolver
        self.args = (odefunc, bounds, rank, device, T, rtol, atol, group)
        n_local = int(bounds[rank + 1] - bounds[rank])
        self.plan = plan if plan is not None else bench_plan(block, bounds, rank, device, group)
        self.local_nnz = self.plan.local_nnz
        self.shard = DeviceShard(self.plan, int(bounds[-1]), group)
        self.solver = DeviceSolver(odefunc, n_local, 'dopri5', rtol, atol, shard=self.shard)
        self.x0 = torch.rand(n_local, odefunc.hidden_size, generator=torch.Generator().manual_seed(rank)).to(device)
        self.out = torch.empty_like(self.x0)
        self.T = T
        self.nfe_done = 0
        self.solver.begin(self.x0, 0.0, borrow=True)

    def run_steps(self, k):
        done = 0
        while done < k:
            before = self.solver.stats()['steps']
            reached = self.solver.advance(self.T, self.out, step_budget=k - done)
            done += int(self.solver.stats()['steps'] - before)
            if reached:
                self.nfe_done += int(self.solver.stats()['nfe'])
                self.solver.begin(self.x0, 0.0, borrow=True)
        return done

    def nfe(self):
        return self.nfe_done + int(self.solver.stats()['nfe'])

    def python_twin(self):
        """The same shard stepped from Python (ShardedBench on the same HaloPlan): bench.py's instrumented pass measures
        the exchange / exposure times there (its streams and launches are the ones this solver issues)."""
        odefunc, bounds, rank, device, T, rtol, atol, group = self.args
        return ShardedBench(odefunc, None, bounds, rank, device, T, rtol, atol, group=group, plan=self.plan)


def bench_plan(block, bounds, rank, device, group=None):
    """The HaloPlan of a bench shard (collective on `group`: every rank builds its plan at the same time), honouring the
    NDCN_SELF_HALO test hook."""
    sh = os.environ.get('NDCN_SELF_HALO', '0')
    return HaloPlan(block, bounds, rank, device, group, self_halo=sh if sh.startswith('scatter:') else int(sh))


def rccl_usable():
    """A rank-LOCAL fact (no collective): can this process bind librccl and draw a unique id?  What bench.py all-reduces
    before any rank enters a collective of the device-resident sharded path."""
    import ctypes
    from . import _lib
    try:
        return _lib.load().ndcn_comm_unique_id(ctypes.create_string_buffer(128)) == 0
    except Exception:
        return False


def grid_row_block(S, world, rank):
    """(row block, bounds) of the metric's weak-scaling grid: the (S * world) x S lattice, rank r owns lattice rows
    [r S, (r + 1) S) and builds only its own rows."""
    from . import graphs
    return graphs.grid_operator_row_block(S * world, S, rank * S, (rank + 1) * S, 'norm_lap'), [r * S * S for r in range(world + 1)]


class ShardedBench:
    """bench.py's N > 1 workload on ANY row-sharded operator: rank r owns rows [bounds[r], bounds[r+1]) of the global
    operator and is handed only that row block (global column indices)."""

    def __init__(self, odefunc, block, bounds, rank, device, T, rtol, atol, ops=None, group=None, plan=None):
        from .torchdiffeq._impl import core
        if ops is None:
            from .ops import hip as ops
        n_local = int(bounds[rank + 1] - bounds[rank])
        self.plan = plan if plan is not None else bench_plan(block, bounds, rank, device, group)
        self.local_nnz = self.plan.local_nnz
        self.func = ShardedODEFunc(odefunc, self.plan, ops)
        self.dops = DistOps(ops, int(bounds[-1]), n_local, group)
        self.x0 = torch.rand(n_local, odefunc.hidden_size, generator=torch.Generator().manual_seed(rank)).to(device)
        self.T, self.rtol, self.atol = T, rtol, atol
        self.core = core
        self._begin()

    def _begin(self):
        f = lambda t, y: (self.func(t, y[0]),)
        self.solver = self.core.Dopri5(self.dops, f, (self.x0,), self.rtol, self.atol, autonomous=True,
                                       fused=self.func)
        self.solver.begin(0.0)

    def run_steps(self, k):
        done = 0
        while done < k:
            before = len(self.solver.log)
            out = self.solver.advance(self.T, step_budget=k - done)
            done += len(self.solver.log) - before
            if out is not None:
                self._nfe_base = getattr(self, '_nfe_base', 0) + self.solver.nfe
                self._begin()
        return done

    def nfe(self):
        return getattr(self, '_nfe_base', 0) + self.solver.nfe


class ShardedGridBench(ShardedBench):
    """the metric's grid, weak scaling: the (S*world) x S grid, rank r owns lattice rows [r*S, (r+1)*S) and builds only
    its own rows (graphs.grid_operator_row_block)."""

    def __init__(self, odefunc, S, world, rank, device, T, rtol, atol, ops=None, group=None):
        from . import graphs
        block = graphs.grid_operator_row_block(S * world, S, rank * S, (rank + 1) * S, 'norm_lap')
        bounds = [r * S * S for r in range(world + 1)]
        super().__init__(odefunc, block, bounds, rank, device, T, rtol, atol, ops, group)
