"""`odeint(func, y0, t, rtol, atol, method, options)` - drop-in for the reference's vendored
torchdiffeq entry point (torchdiffeq/_impl/odeint.py:20-76), computed by HIP kernels.

Two execution paths, both on the GPU:
  * device-resident: `func` is this package's ODEFunc acting on one N x H fp32 panel -> the whole solve
    runs inside libndcn_hip.so (`ndcn_solver_*`): state, stages and dense-output coefficients never
    leave HBM and the host sees one 16-byte record per adaptive step;
  * generic: any callable / tuple state -> the reference's solver control flow (core.py) with one fused
    HIP kernel per bookkeeping chain and `func` called back in Python.
Host tensors are refused: there is no CPU fallback.
"""
import collections
import ctypes
import os
import weakref

import torch

from ... import _lib
from ...ops import hip, new_solve_epoch
from . import core

SOLVERS = {m: m for m in core.METHODS}      # the in-scope subset of odeint.py:8-17
GRAPH_MAX_ELEMS = 1 << 23                   # below ~8M state elements (Pubmed x 256) a step is launch-bound


def _autonomous(func):
    return bool(getattr(func, 'ndcn_autonomous', False))


def _needs_grad(func, y0, probe=None):
    """(needs_grad, probe output or None).  True when the solve must be differentiable: the state or the parameters of
    an nn.Module `func` require grad - or `func` is a plain callable (lambda, bound method, closure over parameters)
    whose output carries autograd history: the reference differentiates through any callable, so that case is detected
    by evaluating `probe()` = func(t0, y0).  That evaluation is the solver's own first one (f0 of dopri5.py:78, k1 of
    the first fixed-grid step): its output is handed back so that the solve REUSES it instead of evaluating twice
    (user-side evaluation counters and RNG-consuming functions see exactly the reference's sequence of calls)."""
    if not torch.is_grad_enabled():
        return False, None
    if any(y.requires_grad for y in y0):
        return True, None
    if isinstance(func, torch.nn.Module):
        return any(p.requires_grad for p in func.parameters()), None
    if probe is not None:
        out = probe()
        return any(torch.is_tensor(o) and o.requires_grad for o in out), out
    return False, None


def _reuse_first_evaluation(func, y0, out):
    """func, except that its FIRST call - which every solver path makes at (t[0], y0) - returns `out`, the value the
    differentiability probe already computed there."""
    pending = [out]

    def wrapped(t, y):
        o = pending[0]
        if o is not None:
            pending[0] = None
            if len(y) == len(y0) and all(a is b for a, b in zip(y, y0)):
                return o
        return func(t, y)
    return wrapped


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, step_log=None):
    """Integrate dy/dt = func(t, y), y(t[0]) = y0; returns y at every t (first dim), y0 first.  (The body is `_odeint`; this frame
    fetches the time grid to the host once for everything below that asks about it: core.grid_scope.)"""
    with core.grid_scope(t):
        return _odeint(func, y0, t, rtol, atol, method, options, step_log)


def _odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, step_log=None):
    """Integrate dy/dt = func(t, y), y(t[0]) = y0; returns y at every t (first dim), y0 first.

    Same signature, defaults, return layout and exceptions as the reference (odeint.py:20-76):
    TypeError for non-float y0 / t, ValueError for `options` without `method`, KeyError for an unknown
    method, AssertionError for a non-monotone t.  Deviations: dopri5 / adams / euler / midpoint / rk4 are
    provided (tsit5 / explicit_adams / fixed_adams raise NotImplementedError); the state must be float32 on a ROCm device;
    `step_log` (a list) optionally receives the dopri5 per-attempt log.
    """
    user_func = func
    t_user = t
    tensor_input, func, y0, t = core.check_inputs(func, y0, t)
    new_solve_epoch()                            # weights written through `.data` since the last solve are packed afresh

    if options is None:
        options = {}
    elif method is None:
        raise ValueError('cannot supply `options` without specifying `method`')
    if method is None:
        method = 'dopri5'
    if method in core.UNSUPPORTED:
        raise NotImplementedError('method %r of the reference is outside the accelerated path '
                                  '(dopri5, adams, euler, midpoint, rk4 are provided)' % method)
    method = SOLVERS[method]                     # KeyError for an unknown name, as the reference's dict lookup

    for y in y0:
        _lib.require_device(y, 'state y0')
    needs_grad, f0 = _needs_grad(user_func, y0, probe=lambda: func(t[0].to(y0[0].dtype), y0))
    if f0 is not None:
        func = _reuse_first_evaluation(func, y0, f0)
    if needs_grad and method in ('euler', 'midpoint', 'rk4') and _device_resident_ok(user_func, tensor_input, y0, t_user, method, options):
        sol = _small_solve_with_grad(user_func, y0[0], t, method)                            # one launch forward, one backward
        if sol is None:
            sol = _fixed_grid_with_grad(user_func, y0[0], t, method)   # any size: fused launches forward, closed-form sweep backward
        if sol is not None:
            return sol
    if needs_grad:
        from .autograd_path import odeint_with_grad
        plain = method == 'dopri5' and _device_resident_ok(user_func, tensor_input, y0, t_user, method, options) and \
            _small_operator(user_func, y0[0]) is not None
        if plain:
            # one autograd node per solve: the native tape (csrc/tape.hip) runs the launches below and their reverse pass itself
            from . import tape
            if tape.applicable(user_func, y0[0], t_user):
                return tape.solve(user_func, y0[0], t, rtol, atol, options, step_log)
        sol = odeint_with_grad(func, y0, t, rtol, atol, method, options, autonomous=_autonomous(user_func),
                               step_log=step_log, odefunc=user_func if plain else None)
    elif _device_resident_ok(user_func, tensor_input, y0, t_user, method, options):
        return _device_resident(user_func, y0[0], t, rtol, atol, method, options, step_log)
    elif method == 'dopri5':
        sol = core.integrate_dopri5(hip, func, y0, t, rtol, atol, autonomous=_autonomous(user_func),
                                    step_log=step_log, **options)
    elif method == 'adams':
        sol = core.integrate_adams(hip, func, y0, t, rtol, atol, autonomous=_autonomous(user_func),
                                   step_log=step_log, **options)
    else:
        if options:
            raise NotImplementedError('fixed-grid options %s: only the default grid (grid == t) is provided'
                                      % sorted(options))
        sol = core.integrate_fixed(hip, func, y0, t, method, autonomous=_autonomous(user_func))
    out = tuple(torch.stack([s[i] for s in sol]) for i in range(len(y0)))
    return out[0] if tensor_input else out


# ---------------------------------------------------------------------------------------------------
# training on a state that fits one compute unit: the whole Euler solve and its reverse sweep, one launch each
# ---------------------------------------------------------------------------------------------------

class _SmallEulerSolve(torch.autograd.Function):
    """FixedGridODESolver.integrate with Euler (and, round 5, midpoint / RK4 3-8) steps (solvers.py:79-99, fixed_grid.py:7-29,
    rk_common.py:72-78) over ODEFunc, differentiated the
    way the reference's drivers train - plain backpropagation through every step (heat_dynamics.py:313-334) - with
    ndcn_solve_small_f32 / ndcn_solve_small_bwd_f32 (csrc/solve_small.hip): the forward's trajectory IS the saved state."""

    @staticmethod
    def forward(ctx, y0, W, b, csr, flags, dts, method='euler'):
        lib = _lib.load()
        n_ticks = len(dts)
        H = y0.shape[1]
        ctx.method = _lib.METHODS[method]
        out = torch.empty((n_ticks + 1,) + tuple(y0.shape), dtype=torch.float32, device=y0.device)
        out[0].copy_(y0)
        arr = (ctypes.c_float * n_ticks)(*dts)
        no_control = bool(flags & _lib.F_NO_CONTROL)
        Wd = None if no_control else W.detach().contiguous()
        bd = None if (no_control or b is None) else b.detach().contiguous()
        view = csr.view_ref(need_symmetric=True) if csr is not None else ctypes.byref(_lib.empty_csr(y0.shape[0]))
        # Euler on the README shapes: the forward launch keeps S_i = A y_i and K_i of every step for the reverse sweep (two panels per
        # tick instead of a gather and a Linear per tick in backward: ndcn_solve_small_keep_*)
        keep = None
        if ctx.method == _lib.M_EULER and csr is not None and Wd is not None and lib.ndcn_solve_small_keep_supported(view, H, flags):
            keep = torch.empty((n_ticks, 2) + tuple(y0.shape), dtype=torch.float32, device=y0.device)
        with torch.cuda.device(y0.device):
            if keep is not None:
                _lib.check(lib.ndcn_solve_small_keep_f32(view, _lib.ptr(Wd), _lib.ptr(bd), H, flags, _lib.ptr(out[0]), arr, n_ticks,
                                                         _lib.ptr(out[1:]), _lib.ptr(keep), _lib.stream_ptr()))
            else:
                _lib.check(lib.ndcn_solve_small_f32(view, _lib.ptr(Wd), _lib.ptr(bd), H, flags, ctx.method, _lib.ptr(out[0]), arr,
                                                    n_ticks, _lib.ptr(out[1:]), _lib.stream_ptr()))
        ctx.csr, ctx.flags, ctx.dts, ctx.keep = csr, flags, arr, keep
        ctx.has_W, ctx.has_b = Wd is not None, bd is not None
        # (W and b through save_for_backward: an in-place parameter change between forward and backward raises, as it does for
        # every other autograd node, instead of differentiating the wrong weights)
        ctx.save_for_backward(out, *([W] if Wd is not None else []), *([b] if bd is not None else []))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        out, *wb = ctx.saved_tensors
        lib = _lib.load()
        Wd = wb[0].detach().contiguous() if ctx.has_W else None
        bd = wb[1 if ctx.has_W else 0].detach().contiguous() if ctx.has_b else None
        H = out.shape[2]
        g = g.contiguous()
        g_y0 = torch.empty_like(out[0])
        g_W = torch.empty((H, H), dtype=torch.float32, device=out.device) if Wd is not None else None
        g_b = torch.empty((H,), dtype=torch.float32, device=out.device) if Wd is not None else None
        csr = ctx.csr
        view = csr.view_ref(need_symmetric=True) if csr is not None else ctypes.byref(_lib.empty_csr(out.shape[1]))
        view_t = csr.transpose().view_ref() if csr is not None else view
        with torch.cuda.device(out.device):
            if ctx.keep is not None:
                _lib.check(lib.ndcn_solve_small_bwd_keep_f32(view, view_t, _lib.ptr(Wd), _lib.ptr(bd), H, ctx.flags, _lib.ptr(out), _lib.ptr(g),
                                                             ctx.dts, len(ctx.dts), _lib.ptr(ctx.keep), _lib.ptr(g_y0), _lib.ptr(g_W),
                                                             _lib.ptr(g_b), _lib.stream_ptr()))
                ctx.keep = None
            else:
                _lib.check(lib.ndcn_solve_small_bwd_f32(view, view_t, _lib.ptr(Wd), _lib.ptr(bd), H, ctx.flags, ctx.method, _lib.ptr(out),
                                                        _lib.ptr(g), ctx.dts, len(ctx.dts), _lib.ptr(g_y0), _lib.ptr(g_W), _lib.ptr(g_b),
                                                        _lib.stream_ptr()))
        return g_y0, g_W, (g_b if bd is not None else None), None, None, None, None


class _FixedGridSolve(torch.autograd.Function):
    """FixedGridODESolver.integrate (solvers.py:79-99) over ODEFunc at ANY size, differentiated the way the drivers train (plain
    backpropagation through every step, heat_dynamics.py:313-334), on the kernels of the inference path:
      forward   the launches of the device-resident solver - the stage algebra of every step rides in the epilogues of its
                right-hand-side launches (ndcn_rhs_rk_f32: Euler / midpoint 1 launch per evaluation, RK4 4 per step); only the
                trajectory - the output - is kept;
      backward  per step, in reverse: the stages are re-formed from the stored state by the same launches (checkpointing: the
                reference's autograd keeps every stage of every step), then the step's vector-Jacobian products in closed
                form - SpMM, the masked Linear backward (g_S, g_W, g_b in one call), SpMM with A^T with the step size folded
                into its alpha - and the stage recurrences as one linear-combination launch each.
    No autograd graph per operation: the torch `add` / `mul` launches between the kernels are gone (they were 17 % of the
    kernel time of a 100k-node Euler training step)."""

    @staticmethod
    def forward(ctx, y0, W, b, csr, flags, method, dts):
        n_ticks = len(dts)
        out = torch.empty((n_ticks + 1,) + tuple(y0.shape), dtype=torch.float32, device=y0.device)
        out[0].copy_(y0)
        no_graph, no_control = bool(flags & _lib.F_NO_GRAPH), bool(flags & _lib.F_NO_CONTROL)
        ctx.meta = (csr, no_graph, no_control, method, dts)
        for i, dt in enumerate(dts):
            _FixedGridSolve._step(csr, out[i], W, b, no_graph, no_control, method, dt, out[i + 1])
        ctx.save_for_backward(out, W, b)
        return out

    @staticmethod
    def _step(csr, y, W, b, no_graph, no_control, method, dt, out_y, keep=None):
        """one step by fused launches; keep (a list) receives [(stage input, K), ...] for the reverse sweep"""
        kw = dict(no_graph=no_graph, no_control=no_control)
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
        if method == 'euler':
            K, _ = hip.rhs_rk(csr, y, W, b, 'combine', y, [], [dt], out_y=out_y, **kw)              # y + dt k1
            stages = [(y, K)]
        elif method == 'midpoint':
            K1, ym = hip.rhs_rk(csr, y, W, b, 'combine', y, [], [f32(dt / 2.0)], **kw)             # y + k1 dt / 2 (an exact halving)
            K2, _ = hip.rhs_rk(csr, ym, W, b, 'combine', y, [], [dt], out_y=out_y, **kw)           # y + dt k2
            stages = [(y, K1), (ym, K2)]
        else:
            stages, x, ks = [], y, []
            for i in range(4):
                K, nxt = hip.rhs_rk(csr, x, W, b, 'rk4', y, ks, [dt], out_y=out_y if i == 3 else None, **kw)
                stages.append((x, K))
                ks = ks + [K]
                x = nxt
        if keep is not None:
            keep.extend(stages)

    @staticmethod
    def _vjp(csr, u, K, g, W, b, no_graph, no_control, alpha):
        """alpha * J(u)^T g for K = relu(W (A u) + b): (g_u, g_W, g_b) with g_W / g_b UNSCALED (the caller scales the small ones)"""
        gW = gb = None
        if no_control:
            gS = hip.relu_bwd(g, K)
        else:
            S = u if no_graph else hip.spmm(csr, u)
            gS, gW, gb = hip.linear_bwd(g, W, S=S, Y=K)
        gu = hip.scale(gS, alpha) if no_graph else hip.spmm(csr.transpose(), gS, alpha=alpha)
        return gu, gW, gb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        out, W, b = ctx.saved_tensors
        csr, no_graph, no_control, method, dts = ctx.meta
        g = g.contiguous()
        n_ticks = len(dts)
        H = out.shape[2]
        a = g[n_ticks]
        gW_tot = torch.zeros((H, H), dtype=torch.float32, device=out.device) if not no_control else None
        gb_tot = torch.zeros((H,), dtype=torch.float32, device=out.device) if not no_control else None
        vj = lambda u, K, gk, alpha: _FixedGridSolve._vjp(csr, u, K, gk, W, b, no_graph, no_control, alpha)

        def acc(gW, gb, scale):
            if gW is not None:
                gW_tot.add_(gW, alpha=scale)
                gb_tot.add_(gb, alpha=scale)
        for i in range(n_ticks - 1, -1, -1):
            dt = dts[i]
            st = []
            scratch = torch.empty_like(out[0])
            _FixedGridSolve._step(csr, out[i], W, b, no_graph, no_control, method, dt, scratch, keep=st)
            if method == 'euler':                                   # y1 = y + dt k1
                (u1, K1), = st
                gu1, gW, gb = vj(u1, K1, a, dt)
                acc(gW, gb, dt)
                a = hip.lincomb([gu1, g[i]], [1.0, 1.0], y0=a)
            elif method == 'midpoint':                              # ym = y + (dt / 2) k1 ; y1 = y + dt k2
                (u1, K1), (u2, K2) = st
                gu2, gW, gb = vj(u2, K2, a, dt)                     # dL/d ym
                acc(gW, gb, dt)
                gu1, gW, gb = vj(u1, K1, gu2, dt / 2.0)
                acc(gW, gb, dt / 2.0)
                a = hip.lincomb([gu2, gu1, g[i]], [1.0, 1.0, 1.0], y0=a)
            else:                                                   # the 3/8 rule, rk_common.py:72-78
                (u1, K1), (u2, K2), (u3, K3), (u4, K4) = st
                c8 = dt / 8.0
                gu4, gW, gb = vj(u4, K4, a, c8)                     # J4^T (c8 a)
                acc(gW, gb, c8)
                gk3 = hip.lincomb([a, gu4], [3.0 * c8, dt])
                gu3, gW, gb = vj(u3, K3, gk3, 1.0)
                acc(gW, gb, 1.0)
                gk2 = hip.lincomb([a, gu4, gu3], [3.0 * c8, -dt, dt])
                gu2, gW, gb = vj(u2, K2, gk2, 1.0)
                acc(gW, gb, 1.0)
                gk1 = hip.lincomb([a, gu4, gu3, gu2], [c8, dt, -dt / 3.0, dt / 3.0])
                gu1, gW, gb = vj(u1, K1, gk1, 1.0)
                acc(gW, gb, 1.0)
                a = hip.lincomb([gu4, gu3, gu2, gu1, g[i]], [1.0] * 5, y0=a)
        return a, gW_tot, (gb_tot if b is not None else None), None, None, None, None


def _fixed_grid_with_grad(odefunc, y0, t, method):
    """The fused-launch training path of a fixed-grid solve over ODEFunc when the one-launch kernels do not take it (any size)."""
    if t.requires_grad or os.environ.get('NDCN_FIXED_GRID_GRAD', '1') == '0' or t.numel() < 2:
        return None
    op = _small_operator(odefunc, y0)
    if op is None:
        return None
    csr, _, flags = op
    core.assert_increasing(t)
    tt = core.host_grid(t).to(y0.dtype)
    dts = (tt[1:] - tt[:-1]).tolist()
    from . import tape
    sol = tape.fixed_grid(_lib.require_device(y0, 'state y0').contiguous(), odefunc.wt.weight, odefunc.wt.bias, csr, flags, method, dts)
    if sol is not None:
        return sol
    return _FixedGridSolve.apply(_lib.require_device(y0, 'state y0').contiguous(), odefunc.wt.weight, odefunc.wt.bias, csr, flags,
                                 method, dts)


def _small_solve_with_grad(odefunc, y0, t, method='euler'):
    """The one-launch training path when the library supports the shape (H <= 31, the state and three work panels in one
    CU's LDS: the reference's README commands), else None - the caller falls back to the per-step autograd path."""
    from ...csr import as_csr
    if t.requires_grad or os.environ.get('NDCN_SOLVE_SMALL_GRAD', '1') == '0' or t.numel() < 2:
        return None
    lib = _lib.load()
    H = odefunc.hidden_size
    flags = _lib.F_RELU | (_lib.F_NO_GRAPH if odefunc.no_graph else 0) | (_lib.F_NO_CONTROL if odefunc.no_control else 0)
    csr = None
    if not odefunc.no_graph:
        csr = as_csr(odefunc.A)
        if csr.device != y0.device or csr.shape[0] != y0.shape[0]:
            return None
    view = csr.view_ref() if csr is not None else ctypes.byref(_lib.empty_csr(y0.shape[0]))
    if not lib.ndcn_solve_small_supported(view, H, flags, _lib.METHODS[method], 1):
        return None
    if method != 'euler' and os.environ.get('NDCN_SOLVE_SMALL_RK_GRAD', '1') == '0':
        return None
    core.assert_increasing(t)
    tt = core.host_grid(t).to(y0.dtype)                    # solvers.py:81: the grid in the state dtype
    dts = (tt[1:] - tt[:-1]).tolist()
    return _SmallEulerSolve.apply(_lib.require_device(y0, 'state y0').contiguous(), odefunc.wt.weight, odefunc.wt.bias, csr, flags, dts, method)


# ---------------------------------------------------------------------------------------------------
# device-resident path
# ---------------------------------------------------------------------------------------------------

def _device_resident_ok(user_func, tensor_input, y0, t, method, options):
    from ...neural_dynamics import ODEFunc
    if not (tensor_input and type(user_func) is ODEFunc):
        return False
    y = y0[0]
    if y.dim() != 2 or y.shape[1] != user_func.hidden_size:
        return False
    if user_func.training and user_func.dropout > 0:
        return False
    if set(options) - set(core.DOPRI5_OPTIONS) or (method != 'dopri5' and options):
        return False
    if options.get('first_step') is not None:                  # dopri5.py:82: then 0.01 is used; host logic handles it
        return False
    th = core.host_grid(t)
    if bool((th[1:] < th[:-1]).any()):          # decreasing grids go through the generic sign flip
        return False
    return method != 'adams'                    # adams steps through the host logic (core.Adams) over the panel kernels


class DeviceSolver:
    """RAII wrapper of ndcn_solver_* for one (ODEFunc, method) pair; the workspace is a torch allocation."""

    def __init__(self, odefunc, n_rows, method, rtol=1e-7, atol=1e-9, max_num_steps=2 ** 31 - 1, use_graph=False,
                 safety=core.SAFETY, ifactor=core.IFACTOR, dfactor=core.DFACTOR, shard=None):
        """shard: a ndcn_amd.sharding.DeviceShard - this rank's part of a node-range sharded graph; the operator is then
        the shard's (own rows, [own | halo] columns) and `odefunc.A` is ignored."""
        from ...csr import as_csr
        self.lib = _lib.load()
        H = odefunc.hidden_size
        flags = _lib.F_RELU | (_lib.F_NO_GRAPH if odefunc.no_graph else 0) | (_lib.F_NO_CONTROL if odefunc.no_control else 0)
        dev = odefunc.wt.weight.device
        self.csr = None
        if shard is not None:
            csr = shard.operator(H)
            assert csr.shape[0] == n_rows
            view = csr.view()
            self._keep = (csr, shard)
        elif odefunc.no_graph:
            view = _lib.empty_csr(n_rows)
            self._keep = ()
        else:
            csr = as_csr(odefunc.A)
            if csr.device != dev:
                raise _lib.NdcnHipError(_lib.EINVAL, 'operator on %s, weights on %s' % (csr.device, dev))
            assert csr.shape[0] == n_rows, 'operator has %d rows, state has %d' % (csr.shape[0], n_rows)
            csr.ensure_plans(H)
            view = csr.view()
            self._keep = (csr,)
            self.csr = csr
        W = odefunc.wt.weight.detach().contiguous()
        b = odefunc.wt.bias.detach().contiguous() if odefunc.wt.bias is not None else None
        _lib.require_device(W, 'weight')
        self._keep += (W, b)
        self.desc = _lib.SolverDesc(_lib.METHODS[method], H, flags, 1 if use_graph else 0, view,
                                    W.data_ptr(), b.data_ptr() if b is not None else None,
                                    float(rtol), float(atol), int(max_num_steps), float(safety), float(ifactor),
                                    float(dfactor), shard.view_ptr(H) if shard is not None else None)
        self.device = dev
        self.shape = (n_rows, H)
        nbytes = int(self.lib.ndcn_solver_workspace_bytes(ctypes.byref(self.desc)))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.handle = ctypes.c_void_p()
        self._pid = os.getpid()
        with torch.cuda.device(dev):
            _lib.check(self.lib.ndcn_solver_create(ctypes.byref(self.desc), _lib.ptr(self.workspace), nbytes,
                                                   ctypes.byref(self.handle)))

    def begin(self, y0, t0, borrow=False):
        """borrow=True (ndcn_solver_begin_borrowed): dopri5 reads y0 in place instead of copying it - the caller leaves
        the tensor alone until the solve is over and never passes an `out` that overlaps it (this object keeps it alive)."""
        y0 = _lib.require_device(y0, 'state y0').contiguous()
        assert tuple(y0.shape) == self.shape
        self._y0 = y0 if borrow else None
        entry = self.lib.ndcn_solver_begin_borrowed if borrow else self.lib.ndcn_solver_begin
        with torch.cuda.device(self.device):
            _lib.check(entry(self.handle, _lib.ptr(y0), float(t0), _lib.stream_ptr()))

    def advance(self, next_t, out=None, step_budget=0):
        """Returns True when next_t was reached (and `out` written), False when the step budget ran out."""
        with torch.cuda.device(self.device):
            rc = _lib.check(self.lib.ndcn_solver_advance(self.handle, float(next_t), _lib.ptr(out), int(step_budget),
                                                         _lib.stream_ptr()))
        return rc == 0

    def advance_many(self, ticks, out):
        """All `ticks` in one library call; out: (len(ticks), n_rows, H) contiguous."""
        assert out.is_contiguous() and tuple(out.shape) == (len(ticks),) + self.shape
        arr = (ctypes.c_double * len(ticks))(*[float(v) for v in ticks])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ndcn_solver_advance_many(self.handle, arr, len(ticks), _lib.ptr(out), _lib.stream_ptr()))

    def stats(self):
        buf = (ctypes.c_double * 6)()
        _lib.check(self.lib.ndcn_solver_stats(self.handle, buf))
        keys = ('steps', 'accepted', 'nfe', 't1', 'dt_next', 'last_ratio')
        return dict(zip(keys, list(buf)))

    def steplog(self):
        n = int(self.lib.ndcn_solver_steplog(self.handle, None, 0))
        buf = (ctypes.c_double * (5 * max(n, 1)))()
        self.lib.ndcn_solver_steplog(self.handle, buf, n)
        return [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]

    def close(self):
        if self.handle and self._pid == os.getpid():          # (never from a fork()ed copy of this object)
            self.lib.ndcn_solver_destroy(self.handle)
        self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _small_operator(odefunc, y0):
    """(csr or None, ctypes view ref, flags) of an ODEFunc for the ndcn_solve_small_* entry points, or None when the operator
    does not live with the state."""
    from ...csr import as_csr
    flags = _lib.F_RELU | (_lib.F_NO_GRAPH if odefunc.no_graph else 0) | (_lib.F_NO_CONTROL if odefunc.no_control else 0)
    if odefunc.no_graph:
        return None, ctypes.byref(_lib.empty_csr(y0.shape[0])), flags
    csr = as_csr(odefunc.A)
    if csr.device != y0.device or csr.shape[0] != y0.shape[0]:
        return None
    return csr, csr.view_ref(), flags


def _small_solve(odefunc, y0, tt, method):
    """ndcn_solve_small_f32 without a solver object (no workspace, no host synchronisation): the inference counterpart of
    _SmallEulerSolve.  tt: the time grid as Python floats ALREADY rounded to the state dtype (solvers.py:81)."""
    import numpy as np
    op = _small_operator(odefunc, y0)
    if op is None:
        return None
    csr, view, flags = op
    lib = _lib.load()
    H = odefunc.hidden_size
    if not lib.ndcn_solve_small_supported(view, H, flags, _lib.METHODS[method], 0):
        return None
    g = np.asarray(tt, dtype=np.float32)
    dts = (g[1:] - g[:-1]).tolist()
    out = torch.empty((len(tt),) + tuple(y0.shape), dtype=torch.float32, device=y0.device)
    out[0].copy_(y0)
    no_control = bool(flags & _lib.F_NO_CONTROL)
    W = None if no_control else _lib.require_device(odefunc.wt.weight.detach().contiguous(), 'weight')
    b = None if (no_control or odefunc.wt.bias is None) else odefunc.wt.bias.detach().contiguous()
    arr = (ctypes.c_float * len(dts))(*dts)
    with torch.cuda.device(y0.device):
        _lib.check(lib.ndcn_solve_small_f32(view, _lib.ptr(W), _lib.ptr(b), H, flags, _lib.METHODS[method], _lib.ptr(out[0]), arr,
                                            len(dts), _lib.ptr(out[1:]), _lib.stream_ptr()))
    return out


# Reference-sized solves (the README commands: 400 x 20) spend a third of their time CREATING the solver - workspace, streams,
# events and above all the capture + instantiation of the per-step hipGraph (~0.5 ms of a 1.7 ms dopri5 solve, rocprofv3 kernel
# trace) - and every odeint call of a training loop's evaluation pass builds the same one again.  Solvers of launch-bound sizes are
# therefore kept (a handful, least recently used out) and re-begun: ndcn_solver_begin resets the state, the captured graph reads the
# weights, the operator and the workspace through pointers that the key pins.
_SOLVERS = collections.OrderedDict()
_SOLVER_CACHE_MAX_ELEMS = 1 << 20
_SOLVER_CACHE_SIZE = 4


def _cached_solver(odefunc, y0, method, rtol, atol, opt, use_graph):
    """(solver, key or None): a kept solver re-used when everything its captured graph points at is unchanged."""
    def make():
        return DeviceSolver(odefunc, y0.shape[0], method, rtol, atol, opt.get('max_num_steps', 2 ** 31 - 1), use_graph=use_graph,
                            safety=opt.get('safety', core.SAFETY), ifactor=opt.get('ifactor', core.IFACTOR),
                            dfactor=opt.get('dfactor', core.DFACTOR))
    if not use_graph or y0.numel() > _SOLVER_CACHE_MAX_ELEMS or os.environ.get('NDCN_SOLVER_CACHE', '1') == '0':
        return make(), None
    W, b = odefunc.wt.weight, odefunc.wt.bias
    key = (id(odefunc), W.data_ptr(), None if b is None else b.data_ptr(), id(getattr(odefunc, 'A', None)), tuple(y0.shape), method,
           float(rtol), float(atol), tuple(sorted((k, v) for k, v in opt.items() if v is not None)), y0.device.index,
           torch.cuda.current_stream(y0.device).cuda_stream, bool(odefunc.no_graph), bool(odefunc.no_control))
    hit = _SOLVERS.get(key)
    if hit is not None and not odefunc.no_graph:
        # id(A) names an object, not its contents: an in-place write to A re-converts (csr.as_csr keys on A._version), and a new
        # tensor may re-use a freed one's id - the kept solver's captured graph would go on reading the OLD operator arrays that
        # its _keep holds alive.  The CsrOperator the solver was built on must be the one the operator converts to now.
        from ...csr import as_csr
        if getattr(hit[1], 'csr', None) is not as_csr(odefunc.A):
            del _SOLVERS[key]
            if not getattr(hit[1], '_in_use', False):
                hit[1].close()
            hit = None
    if hit is not None and hit[0]() is odefunc and not getattr(hit[1], '_in_use', False) and hit[1].handle:
        _SOLVERS.move_to_end(key)
        hit[1]._in_use = True
        return hit[1], key
    solver = make()
    solver._in_use = True
    if hit is None:
        _SOLVERS[key] = (weakref.ref(odefunc), solver)
        while len(_SOLVERS) > _SOLVER_CACHE_SIZE:
            _, (_, old) = _SOLVERS.popitem(last=False)
            if not getattr(old, '_in_use', False):
                old.close()
        return solver, key
    return solver, None                                      # the kept one is busy (another thread): a solver of its own


def _device_resident(odefunc, y0, t, rtol, atol, method, options, step_log):
    core.assert_increasing(t)
    tt = core.host_grid(t).to(torch.float64).tolist()
    if method != 'dopri5':
        # solvers.py:81: the fixed grid is t in the state dtype
        tt = core.host_grid(t).to(y0.dtype).to(torch.float64).tolist()
    if method != 'dopri5' and len(tt) > 1:
        out = _small_solve(odefunc, y0, tt, method)       # a state that fits one compute unit: the whole grid in ONE launch
        if out is not None:
            return out
    # launch-bound sizes replay ONE captured hipGraph per step - a fixed-grid step, or one attempted dopri5 step - with the
    # step size in device memory (the library declines where a path has no replayable form)
    use_graph = y0.numel() <= GRAPH_MAX_ELEMS and os.environ.get('NDCN_HIPGRAPH', '1') != '0'
    opt = core.dopri5_options(options, 1) if method == 'dopri5' else {}
    solver, cache_key = _cached_solver(odefunc, y0, method, rtol, atol, opt, use_graph)
    try:
        out = torch.empty((len(tt),) + tuple(y0.shape), dtype=torch.float32, device=y0.device)
        out[0].copy_(y0)
        solver.begin(out[0], tt[0], borrow=True)           # the solution's first panel IS the initial state: read in place
        try:
            if len(tt) > 1:
                solver.advance_many(tt[1:], out[1:])               # one library call for the whole time vector
        except _lib.NdcnHipError as e:
            if e.code in (_lib.ENONFINITE, _lib.EUNDERFLOW, _lib.EMAXSTEPS, _lib.ESTATE):
                raise AssertionError(str(e)) from None         # the reference raises AssertionError here
            raise
        if step_log is not None:
            step_log.extend(solver.steplog())
            step_log.append(('nfe', int(solver.stats()['nfe'])))
        torch.cuda.current_stream().synchronize()    # the workspace is released (or handed to the next solve) below
        return out
    finally:
        if cache_key is None:
            solver.close()
        else:
            solver._in_use = False
