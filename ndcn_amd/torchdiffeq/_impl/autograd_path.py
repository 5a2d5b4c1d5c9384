"""odeint when a gradient is required: backpropagation THROUGH the solver, as the reference drivers train
(heat_dynamics.py:333, dgnn.py:204; the reference never enables the adjoint).

Fixed-grid methods: the solver control flow of core.py over differentiable HIP ops (ndcn_amd/autograd_ops.py);
the step sizes are data-independent, so that is the reference's gradient exactly.

dopri5: the reference's (old) torchdiffeq keeps dt, t0/t1, the initial step and the interpolation abscissa as
TENSORS with autograd history - the step-size controller is part of the differentiated graph
(misc.py:84-170, dopri5.py:94-122).  Dropping those paths changes what the model learns (measured on Cora with
the README command: 81.6 % with them, 78.6 % without), so they are kept: the scalar chain below is torch 0-d
tensors exactly as in the reference, every panel VALUE comes from the HIP kernel the inference path uses, and
every panel op's backward is ONE analytic VJP kernel (csrc/rk_bwd.hip: panel gradients + the inner products that are
the gradients of the scalar coefficients).  The RHS itself (`func`) is differentiated by its own autograd Function (HIP SpMM / Linear).
"""
import math
import os
import threading

import numpy as np
import torch

from ...autograd_ops import autograd_ops
from ...ops import hip
from . import core

f32 = np.float32


def odeint_with_grad(func, y0, t, rtol, atol, method, options, autonomous=False, step_log=None, odefunc=None):
    if method == 'dopri5':
        return integrate_dopri5_grad(func, y0, t, rtol, atol, autonomous=autonomous, step_log=step_log, odefunc=odefunc, **options)
    if method == 'adams':
        # gradients flow through every panel operation of the accepted steps; the step sizes and orders the controller
        # chose are constants of the graph (the reference's autograd also follows dt through the error ratios - a
        # deviation that is stated, not hidden: DESIGN section 2)
        return core.integrate_adams(autograd_ops, func, y0, t, rtol, atol, autonomous=autonomous, step_log=step_log, **options)
    if options:
        raise NotImplementedError('fixed-grid options %s: only the default grid (grid == t) is provided' % sorted(options))
    return core.integrate_fixed(autograd_ops, func, y0, t, method, autonomous=autonomous)


class _HipValue(torch.autograd.Function):
    """forward: the value a HIP kernel computes; backward: autograd through `torch_fn`, an equivalent torch
    expression of the same inputs (device panels and CPU 0-d scalars alike), recomputed on demand."""

    @staticmethod
    def forward(ctx, hip_fn, torch_fn, *inputs):
        ctx.torch_fn = torch_fn
        ctx.save_for_backward(*inputs)
        with torch.no_grad():
            return hip_fn(*[i.detach() for i in inputs])

    @staticmethod
    def backward(ctx, g):
        inputs = ctx.saved_tensors
        needs = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            xs = [i.detach().requires_grad_(bool(n)) for i, n in zip(inputs, needs)]
            out = ctx.torch_fn(*xs)
            wanted = [x for x in xs if x.requires_grad]
            grads = torch.autograd.grad(out, wanted, g.to(out.device, out.dtype), allow_unused=True) if wanted else ()
        it = iter(grads)
        res = []
        for x in xs:
            if x.requires_grad:
                gx = next(it)
                res.append(torch.zeros_like(x) if gx is None else gx)
            else:
                res.append(None)
        return (None, None) + tuple(res)


def _on(dev, s):
    return s.to(dev) if s.device != dev else s


# ---- the panel ops: HIP forward, HIP vector-Jacobian product (csrc/rk_bwd.hip) --------------------------------------
# NDCN_VJP=torch switches every op below back to `_HipValue` (autograd through the torch expression): the A/B the GPU
# tests use to check the analytic kernels.

def _analytic():
    return os.environ.get('NDCN_VJP', 'hip') != 'torch'


def _scalar_like(ref, v):
    return torch.tensor(v, dtype=ref.dtype, device=ref.device)


def _active(ks, cs):
    """the terms the forward kernels are handed: zero coefficients are dropped (misc.py:22-25 adds exact zeros)"""
    idx = [j for j, c in enumerate(cs) if float(c) != 0.0]
    return idx, [ks[j] for j in idx], [f32(float(cs[j])) for j in idx]


class _CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, y0, *rest):
        ks, cs = rest[:n], rest[n:]
        idx, kk, cc = _active(ks, cs)
        ctx.n, ctx.idx, ctx.cc = n, idx, cc
        ctx.save_for_backward(*kk, *cs)
        return hip.combine(y0, kk, cc) if kk else y0.clone()

    @staticmethod
    def backward(ctx, g):
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        kk, cs = saved[:len(idx)], saved[len(idx):]
        needs = ctx.needs_input_grad
        need_k, need_c = needs[2:2 + n], needs[2 + n:]
        gk_all, gc_all = [None] * n, [None] * n
        if idx and (any(need_k[j] for j in idx) or any(need_c[j] for j in idx)):
            gk, dots = hip.combine_bwd(g, kk, cc, [need_k[j] for j in idx], need_dots=any(need_c[j] for j in idx))
            for q, j in enumerate(idx):
                gk_all[j] = gk[q]
                if need_c[j]:
                    gc_all[j] = _scalar_like(cs[j], dots[q])
        return (None, g if needs[1] else None) + tuple(gk_all) + tuple(gc_all)


def _combine(y0, ks, cs):
    """y0 + sum_j c_j k_j with c_j 0-d tensors (dt * beta in the state dtype, misc.py:22-25)."""
    n = len(ks)
    if _analytic():
        return _CombineFn.apply(n, y0, *ks, *cs)

    def hip_fn(y, *rest):
        _, kk, cc = _active(rest[:n], rest[n:])
        return hip.combine(y, kk, cc) if kk else y.clone()

    def torch_fn(y, *rest):
        acc = 0
        for k, c in zip(rest[:n], rest[n:]):
            acc = acc + _on(y.device, c) * k
        return y + acc

    return _HipValue.apply(hip_fn, torch_fn, y0, *ks, *cs)


class _ErrorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, rtol, atol, bad_out, y0, y1, *rest):
        ks, cs = rest[:n], rest[n:]
        idx, kk, cc = _active(ks, cs)
        s, bad = hip.error(y0, y1, kk, cc, rtol, atol)
        bad_out.append(bad)
        ctx.n, ctx.idx, ctx.cc, ctx.tol = n, idx, cc, (rtol, atol)
        ctx.save_for_backward(y0, y1, *kk, *cs)
        return torch.tensor(f32(s / y0.numel()), dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        y0, y1 = saved[:2]
        kk, cs = saved[2:2 + len(idx)], saved[2 + len(idx):]
        needs = ctx.needs_input_grad
        need_k, need_c = needs[6:6 + n], needs[6 + n:]
        g_r = float(g)
        gy0, gy1, gk, dots = hip.error_bwd(y0, y1, kk, cc, ctx.tol[0], ctx.tol[1], g_r, needs[4], needs[5],
                                           [need_k[j] for j in idx], need_dots=any(need_c[j] for j in idx))
        gk_all, gc_all = [None] * n, [None] * n
        for q, j in enumerate(idx):
            gk_all[j] = gk[q]
            if need_c[j]:
                gc_all[j] = _scalar_like(cs[j], g_r * dots[q])
        return (None, None, None, None, gy0, gy1) + tuple(gk_all) + tuple(gc_all)


def _error_ratio(y0, y1, ks, cs, rtol, atol, bad_out):
    """mean(((sum_j c_j k_j) / (atol + rtol max(|y0|, |y1|)))^2) as a float32 0-d CPU tensor (misc.py:146-157)."""
    n = len(ks)
    if _analytic():
        return _ErrorFn.apply(n, rtol, atol, bad_out, y0, y1, *ks, *cs)

    def hip_fn(a, b, *rest):
        _, kk, cc = _active(rest[:n], rest[n:])
        s, bad = hip.error(a, b, kk, cc, rtol, atol)
        bad_out.append(bad)
        return torch.tensor(f32(s / a.numel()), dtype=torch.float32)

    def torch_fn(a, b, *rest):
        e = 0
        for k, c in zip(rest[:n], rest[n:]):
            e = e + _on(a.device, c) * k
        tol = atol + rtol * torch.max(torch.abs(a), torch.abs(b))
        r = e / tol
        return torch.mean(r * r)

    return _HipValue.apply(hip_fn, torch_fn, y0, y1, *ks, *cs)


def _rms_value(s, numel):
    nrm = f32(math.sqrt(s)) if s == s and s >= 0 else f32('nan')
    return torch.tensor(f32(nrm / f32(math.sqrt(numel))), dtype=torch.float32)


class _RmsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rtol, atol, bad_out, has_b, *xs):
        a, y = xs[0], xs[-1]
        b = xs[1] if has_b else None
        s, bad = hip.scaled_sumsq(a, b, y, rtol, atol)
        if bad_out is not None:
            bad_out.append(bad)
        ctx.has_b, ctx.tol, ctx.s = has_b, (rtol, atol), s
        ctx.save_for_backward(*xs)
        return _rms_value(s, a.numel())

    @staticmethod
    def backward(ctx, g):
        xs = ctx.saved_tensors
        a, y = xs[0], xs[-1]
        b = xs[1] if ctx.has_b else None
        needs = ctx.needs_input_grad[4:]
        # d (||v|| / sqrt(N)) / d v = v / (||v|| sqrt(N)); torch's norm backward is 0 at v = 0
        coef = float(g) / (math.sqrt(ctx.s) * math.sqrt(a.numel())) if ctx.s > 0 else 0.0
        ga, gb, gy = hip.rms_bwd(a, b, y, ctx.tol[0], ctx.tol[1], coef, needs[0], ctx.has_b and needs[1], needs[-1])
        return (None, None, None, None) + ((ga, gb, gy) if ctx.has_b else (ga, gy))


def _rms(a, b, y, rtol, atol, bad_out=None):
    """misc.py:71-76 of (a [- b]) / (atol + |y| rtol): float32 0-d CPU tensor."""
    has_b = b is not None
    args = (a, b, y) if has_b else (a, y)
    if _analytic():
        return _RmsFn.apply(rtol, atol, bad_out, has_b, *args)

    def hip_fn(*xs):
        aa, yy = xs[0], xs[-1]
        bb = xs[1] if has_b else None
        s, bad = hip.scaled_sumsq(aa, bb, yy, rtol, atol)
        if bad_out is not None:
            bad_out.append(bad)
        return _rms_value(s, aa.numel())

    def torch_fn(*xs):
        aa, yy = xs[0], xs[-1]
        scale = atol + torch.abs(yy) * rtol
        v = ((aa - xs[1]) if has_b else aa) / scale
        return v.norm() / (v.numel() ** 0.5)

    return _HipValue.apply(hip_fn, torch_fn, *args)


def _dense_value(cache, a0, a1, kk, dt_, x_):
    dt32 = f32(float(dt_))
    if 'fit' not in cache:
        cmid = [f32(dt32 * f32(c)) for c in core.DP_C_MID]
        cache['fit'] = hip.interp_fit(a0, a1, list(kk), cmid, dt32)
    xv = f32(float(x_))
    x2 = f32(xv * xv)
    x3 = f32(x2 * xv)
    x4 = f32(x3 * xv)
    return hip.interp_eval(cache['fit'], a0, (x4, x3, x2, xv, f32(1)))


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cache, a0, a1, *rest):
        kk, dt_, x_ = rest[:7], rest[7], rest[8]
        ctx.save_for_backward(a0, a1, *rest)
        return _dense_value(cache, a0, a1, kk, dt_, x_)

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        a0, a1, kk, dt_, x_ = saved[0], saved[1], saved[2:9], saved[9], saved[10]
        needs = ctx.needs_input_grad
        gy0, gy1, gk, d_x, d_dt = hip.interp_bwd(g, a0, a1, list(kk), f32(float(dt_)), f32(float(x_)), needs[1], needs[2],
                                                 list(needs[3:10]), need_dots=needs[10] or needs[11])
        return (None, gy0, gy1) + tuple(gk) + (_scalar_like(dt_, d_dt) if needs[10] else None,
                                               _scalar_like(x_, d_x) if needs[11] else None)


def _dense_output(y0, y1, ks, dts, x, cache):
    """dopri5.py:39-45 + interp.py:21-35,58-65 at abscissa x (0-d, state dtype) for step size dts (0-d)."""
    if _analytic():
        return _DenseFn.apply(cache, y0, y1, *ks, dts, x)

    def hip_fn(a0, a1, *rest):
        return _dense_value(cache, a0, a1, rest[:7], rest[7], rest[8])

    def torch_fn(a0, a1, *rest):
        kk = rest[:7]
        dt_, x_ = _on(a0.device, rest[7]), _on(a0.device, rest[8])
        ym = 0
        for k, c in zip(kk, core.DP_C_MID):
            ym = ym + (dt_ * c) * k
        ym = a0 + ym
        f0, f1 = kk[0], kk[6]
        ca = (-2 * dt_) * f0 + (2 * dt_) * f1 + -8 * a0 + -8 * a1 + 16 * ym
        cb = (5 * dt_) * f0 + (-3 * dt_) * f1 + 18 * a0 + 14 * a1 + -32 * ym
        cc = (-4 * dt_) * f0 + dt_ * f1 + -11 * a0 + -5 * a1 + 16 * ym
        cd = dt_ * f0
        x2 = x_ * x_
        x3 = x2 * x_
        x4 = x3 * x_
        return ca * x4 + cb * x3 + cc * x2 + cd * x_ + a0

    return _HipValue.apply(hip_fn, torch_fn, y0, y1, *ks, dts, x)


# ---- "carry" forms of the three panel operations ---------------------------------------------------------------------
# Every stage derivative of a Runge-Kutta step feeds SEVERAL later operations (the following stage sums, the error estimate,
# the dense output), and so does the step's initial state.  Autograd adds the gradients such a tensor receives with one
# elementwise pass per consumer: 258 `add` launches = 24 % of the kernel time of a dopri5 training step on the 100k-node grid
# (rocprofv3, tools/prof_train.py).  The carry forms hand their panel operands on as extra OUTPUTS (aliases), and the solver
# below threads them from operation to operation, so that every panel has ONE consumer; the gradient a panel has received
# from the later operations then arrives as the gradient of the carried output, and the VJP kernel adds its own contribution
# in the pass it makes anyway (the `acc` operands of csrc/rk_bwd.hip).  Same values, same graph semantics - the sums are
# formed in a different order than autograd's, to fp32 rounding.  NDCN_GRAD_CARRY=0 restores the fan-out forms (A/B test).

def _carry():
    return _analytic() and os.environ.get('NDCN_GRAD_CARRY', '1') != '0'


# ---- deferred scalar gradients -----------------------------------------------------------------------------------------
# The VJP kernels reduce the gradients of their SCALAR inputs (dt * beta, the error coefficients, the interpolation abscissa) on
# the device; reading each of them back where it is produced stops the host ~35 times per training step with nothing queued
# behind (17 % idle GPU on the 100k-node case).  In carry mode the kernels' sums come back as DEFERRED host tensors (ops.
# _BwdDots.lazy: a pinned slot an asynchronous copy fills, plus an event) and every scalar input of a carry op passes through
# `_Await` - an identity whose backward waits for that event before anything reads the value.  The `_Await` nodes take host
# tensors, so autograd runs them on the thread that called backward(), while the panel nodes run on the device's worker thread:
# the wait blocks the scalar chain only, the worker keeps queueing panel launches.  NDCN_GRAD_LAZY=0: eager reads.

def _lazy():
    return os.environ.get('NDCN_GRAD_LAZY', '1') != '0'


class _LazyFlag(threading.local):
    """Do the carry ops of the solve THIS THREAD is building defer their scalars?  Per thread (a thread per device builds its own
    solve), saved and restored around a solve (a right-hand side that solves an inner problem): round-4 advisor."""
    on = False
    keep_s = False                          # do the fused evaluations keep S = A x for the weight gradient? (_keep_s_enabled)


_LAZY = _LazyFlag()


class _Await(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from ...ops import _BwdDots
        return _BwdDots.await_lazy(g)


def _grad_scalar(ref, v):
    """gradient of a scalar input: a deferred 0-d tensor as it is (float32, like every coefficient), a float through torch.tensor"""
    return v if torch.is_tensor(v) else torch.tensor(v, dtype=ref.dtype, device=ref.device)


class _StageCarryFn(torch.autograd.Function):
    """(u, y0', k_1', ..) = (y0 + sum_j c_j k_j, y0, k_1, ..)"""

    @staticmethod
    def forward(ctx, n, y0, *rest):
        ks, cs = rest[:n], rest[n:]
        idx, kk, cc = _active(ks, cs)
        ctx.n, ctx.idx, ctx.cc = n, idx, cc
        ctx.lazy = _LAZY.on                  # (the solver below routes every coefficient of a carry op through _Await)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*kk, *cs)
        u = hip.combine(y0, kk, cc) if kk else y0.clone()
        return (u, y0) + tuple(ks)

    @staticmethod
    def backward(ctx, g, g_y0c, *g_kc):
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        kk, cs = saved[:len(idx)], saved[len(idx):]
        needs = ctx.needs_input_grad
        need_y0, need_k, need_c = needs[1], needs[2:2 + n], needs[2 + n:]
        gk_all = [g_kc[j] if need_k[j] else None for j in range(n)]
        gc_all = [None] * n
        g_y0 = g_y0c if need_y0 else None
        if g is not None:
            g = g.contiguous()
            want_dots = any(need_c[j] for j in idx)
            acc_y0 = g_y0c if (need_y0 and g_y0c is not None) else None
            if idx and (any(need_k[j] for j in idx) or want_dots or acc_y0 is not None):
                res = hip.combine_bwd(g, kk, cc, [need_k[j] for j in idx], need_dots=want_dots,
                                      accs=[g_kc[j] if need_k[j] else None for j in idx], acc_y0=acc_y0, lazy=ctx.lazy)
                gk, dots = res[0], res[1]
                for q, j in enumerate(idx):
                    if need_k[j]:
                        gk_all[j] = gk[q]
                    if need_c[j]:
                        gc_all[j] = _grad_scalar(cs[j], dots[q])
                if need_y0:
                    g_y0 = res[2] if acc_y0 is not None else g
            elif need_y0:
                g_y0 = g if g_y0c is None else hip.combine(g_y0c.contiguous(), [g], [f32(1)])
        return (None, g_y0) + tuple(gk_all) + tuple(gc_all)


# ---- a fixed summation order for tensors with SEVERAL consumers (round 5: same seed -> same bits) ---------------------------------
# The scalar chain of the step-size controller lives on the host, the panels on the device: autograd runs the two kinds of nodes on
# two threads, so WHEN a panel node becomes ready depends on the host thread's progress, and with it the order in which the engine
# adds the gradients a tensor receives from three or more consumers - fp32 sums in a different order, one ulp apart, once every
# ~10-20 training steps (found with tools/bench_dgnn.py: the first tensor that differed between two runs of one seed was the gradient
# leaving the ODE block, at epoch 9 resp. 18, by 5e-10 - with h, the solve, the logits, the loss and the incoming gradient still
# bit-equal).  What the carry forms leave with several consumers gets ONE consumer here:
#   _FanOut      the solve's y0 (first evaluation, the three norms and the probe step of the initial-step selection, the first stage
#                chain, the trajectory's first tick) and f0 inside the initial-step selection: n aliases, their gradients added left to
#                right by one ndcn_rk_combine_f32 launch;
#   W, b         ride through the evaluations of the fused path like the panels do (carried outputs): evaluation i adds its g_W to what
#                evaluations i + 1 .. sent - a chain the data dependencies order, whatever the threads do.

class _FanOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g.contiguous() for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        if not live[0].is_cuda:                                      # scalars of the controller chain: left to right on the host
            tot = live[0]
            for g in live[1:]:
                tot = tot + g
            return tot, None
        return hip.lincomb(live, [1.0] * len(live)), None


def _fan(x, n):
    return _FanOut.apply(x, n) if (n > 2 and x.requires_grad) else (x,) * n


# The ~28 products dt * beta_ij, dt * c_err_j of one attempted step as ONE multiplication by the tableau (bit-identical products:
# float32 x float32 either way) and one unbind: `dts` then has one consumer instead of 28 whose gradients - each born on the device
# thread, each a float32 scalar - the engine would add in arrival order.
_TABLEAU = None


def _step_coefficients(dts):
    """-> (rows, c_err): rows[i][j] = dts * DP_BETA[i][j], c_err[j] = dts * DP_C_ERR[j] as 0-d float32 tensors"""
    global _TABLEAU
    if _TABLEAU is None:
        flat = [b for row in core.DP_BETA for b in row] + list(core.DP_C_ERR)
        _TABLEAU = torch.tensor(flat, dtype=torch.float32)
    prod = (dts * _TABLEAU.to(dts.dtype)).unbind(0)
    rows, o = [], 0
    for row in core.DP_BETA:
        rows.append(list(prod[o:o + len(row)]))
        o += len(row)
    return rows, list(prod[o:o + len(core.DP_C_ERR)])


def _add_carried(own, carried):
    """own + carried for the small parameter gradients of the chain (either may be None)"""
    if own is None:
        return carried
    return own if carried is None else own + carried


class _RhsCarryFn(torch.autograd.Function):
    """(K, W', b') = (relu(W (A u) + b), W, b): a plain evaluation of the fused path with the parameters handed on (see above)."""

    @staticmethod
    def forward(ctx, op, u, W, b):
        A, no_graph, no_control = op
        K = hip.rhs(A, u, W, b, no_graph=no_graph, no_control=no_control)
        ctx.op, ctx.has_b = op, b is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(u, W, K)
        return K, W.view_as(W), (b.view_as(b) if b is not None else None)

    @staticmethod
    def backward(ctx, g_K, g_Wc, g_bc):
        from ...autograd_ops import rhs_vjp
        u, W, K = ctx.saved_tensors
        needs = ctx.needs_input_grad                                  # (op, u, W, b)
        gu = gW = gb = None
        if g_K is not None:
            A, no_graph, no_control = ctx.op
            gu, gW, gb = rhs_vjp(A, no_graph, no_control, u, W, K, g_K.contiguous(), needs[1], needs[2], ctx.has_b and needs[3])
        return None, gu, _add_carried(gW, g_Wc) if needs[2] else None, _add_carried(gb, g_bc) if (ctx.has_b and needs[3]) else None


class _RhsStageCarryFn(torch.autograd.Function):
    """(K, u', y0', W', b', k_1', ..) = (relu(W (A u) + b), y0 + sum_j c_j k_j + c_new K, y0, k_1, ..): one evaluation of ODEFunc and the
    NEXT stage input in ONE launch (ndcn_rhs_rk_f32, mode combine - the launch of the inference solver), i.e. `_Rhs` followed by
    `_StageCarryFn` as one node: the forward saves the combine launch (5 of a dopri5 step's 6), the backward runs the same two
    VJPs - the combine's first (its gradient for K joins what K's later consumers sent), then the right-hand side's."""

    @staticmethod
    def forward(ctx, n, op, u, W, b, y0, *rest):
        ks, cs = rest[:n], rest[n:]                                   # n earlier stages, n + 1 coefficients (the new K last)
        A, no_graph, no_control = op
        idx, kk, cc = _active(ks, cs[:n])
        c_new = f32(float(cs[n]))
        K, u_next = hip.rhs_rk(A, u, W, b, 'combine', y0, kk, cc + [c_new], no_graph=no_graph, no_control=no_control)
        ctx.n, ctx.idx, ctx.cc, ctx.op, ctx.has_b = n, idx, cc + [c_new], op, b is not None
        ctx.lazy = _LAZY.on
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(u, W, K, *kk, *cs)
        return (K, u_next, y0, W.view_as(W), (b.view_as(b) if b is not None else None)) + tuple(ks)

    @staticmethod
    def backward(ctx, g_K, g_u, g_y0c, g_Wc, g_bc, *g_kc):
        from ...autograd_ops import rhs_vjp
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        u, W, K = saved[:3]
        kk, cs = saved[3:3 + len(idx)], saved[3 + len(idx):]
        needs = ctx.needs_input_grad                                  # (n, op, u, W, b, y0, k.., c..)
        need_u, need_w, need_b, need_y0 = needs[2], needs[3], ctx.has_b and needs[4], needs[5]
        need_k, need_c = needs[6:6 + n], needs[6 + n:]
        gk_all = [g_kc[j] if need_k[j] else None for j in range(n)]
        gc_all = [None] * (n + 1)
        g_y0 = g_y0c if need_y0 else None
        if g_u is not None:
            g_u = g_u.contiguous()
            acc_y0 = g_y0c if (need_y0 and g_y0c is not None) else None
            want_dots = any(need_c[j] for j in idx) or need_c[n]
            res = hip.combine_bwd(g_u, list(kk) + [K], cc, [need_k[j] for j in idx] + [True], need_dots=want_dots,
                                  accs=[g_kc[j] if need_k[j] else None for j in idx] + [g_K], acc_y0=acc_y0, lazy=ctx.lazy)
            gk, dots = res[0], res[1]
            for q, j in enumerate(idx):
                if need_k[j]:
                    gk_all[j] = gk[q]
                if need_c[j]:
                    gc_all[j] = _grad_scalar(cs[j], dots[q])
            if need_c[n]:
                gc_all[n] = _grad_scalar(cs[n], dots[len(idx)])
            g_K = gk[len(idx)]
            if need_y0:
                g_y0 = res[2] if acc_y0 is not None else g_u
        gu_in = gW = gb = None
        if g_K is not None:
            A, no_graph, no_control = ctx.op
            gu_in, gW, gb = rhs_vjp(A, no_graph, no_control, u, W, K, g_K.contiguous(), need_u, need_w, need_b)
        gW = _add_carried(gW, g_Wc) if need_w else None
        gb = _add_carried(gb, g_bc) if need_b else None
        return (None, None, gu_in, gW, gb, g_y0) + tuple(gk_all) + tuple(gc_all)


# ---- the stage sums' VJPs in PULL form (round 5) ---------------------------------------------------------------------------------
# In the carry forms above every stage's backward PUSHES its contribution c_mj g_u_m into the received gradient of every earlier k_j
# (and of y0): one read-modify-write per (stage, earlier stage) pair - 88 panels per attempted step, `combine_bwd` 21-29 % of the
# kernel time of a dopri5 training step.  Here a stage's backward only NOTES (c_mj, g_u_m) in the step's `_StepPull`; the node that
# produced k_j adds up what was noted for it when its own turn comes (one ndcn_rk_combine_f32 over <= 7 panels, left to right: a fixed
# order), and the first node of the step does the same for y0 and k_1.  The inner products <g_u_m, k_j> that become the coefficients'
# gradients still cost one read of the k panels per stage (ndcn_rk_combine_bwd_f32 without outputs).  67 panels instead of 88.
# The data dependencies of the carried tensors order the nodes of a step whatever autograd's threads do, so the notes are complete
# when they are read.  NDCN_GRAD_PULL=0: the push forms.

class _StepPull:
    def __init__(self):
        self.for_k = {}                                              # k index within the step -> [(c, g_u), ...]
        self.for_y0 = []

    def note(self, j, c, g):
        self.for_k.setdefault(j, []).append((c, g))

    def take(self, j, received):
        """received (may be None) + sum of the noted c g_u for k_j, added left to right in one launch"""
        terms = self.for_k.pop(j, [])
        panels = ([received.contiguous()] if received is not None else []) + [g for _, g in terms]
        coefs = ([f32(1)] if received is not None else []) + [c for c, _ in terms]
        if not panels:
            return None
        if len(panels) == 1 and float(coefs[0]) == 1.0:
            return panels[0]
        return hip.lincomb(panels, coefs)

    def take_y0(self, received):
        panels = ([received.contiguous()] if received is not None else []) + self.for_y0
        self.for_y0 = []
        if not panels:
            return None
        return panels[0] if len(panels) == 1 else hip.lincomb(panels, [f32(1)] * len(panels))


def _pull_enabled():
    return os.environ.get('NDCN_GRAD_PULL', '1') != '0'


def _keep_s(op, mode, n_prev, x):
    """a panel for S = A x when the fused launch can write it on the side (ndcn_rhs_rk_adj_f32) and the solve keeps it for the weight
    gradient (one panel per evaluation held until backward instead of one SpMM per evaluation in backward), else None"""
    A, no_graph, no_control = op
    if not getattr(_LAZY, 'keep_s', False) or no_graph or no_control or x.shape[1] != 256:
        return None
    return torch.empty_like(x) if hip.rhs_adj_supported(A, 256, mode, n_prev) else None


def _keep_s_enabled(y):
    """NDCN_GRAD_KEEP_S = 1 / 0 / auto (default): keep S when ~60 more panels (a long solve's evaluations) fit the free device memory
    four times over; otherwise the SpMM is recomputed in backward as before"""
    v = os.environ.get('NDCN_GRAD_KEEP_S', 'auto')
    if v in ('0', '1'):
        return v == '1'
    free, _ = torch.cuda.mem_get_info(y.device)
    return 240 * y.numel() * 4 <= free


def _row_dot_enabled():
    return os.environ.get('NDCN_GRAD_ROW_DOT', '1') != '0'


class _StagePullFn(torch.autograd.Function):
    """_StageCarryFn with one earlier stage - the first node of a fused step: (u, y0', k_1') = (y0 + c k_1, y0, k_1) - in pull form:
    its backward runs LAST in the step and hands y0 and k_1 everything the step noted for them."""

    @staticmethod
    def forward(ctx, pull, y0, k1, c):
        cc = f32(float(c))
        ctx.pull, ctx.cc, ctx.lazy = pull, cc, _LAZY.on
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(k1, c)
        return hip.combine(y0, [k1], [cc]), y0, k1

    @staticmethod
    def backward(ctx, g, g_y0c, g_k1c):
        k1, c = ctx.saved_tensors
        need_y0, need_k, need_c = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        pull, gc = ctx.pull, None
        if g is not None:
            g = g.contiguous()
            if need_c:
                res = hip.combine_bwd(g, [k1], [ctx.cc], [False], need_dots=True, lazy=ctx.lazy)
                gc = _grad_scalar(c, res[1][0])
            pull.note(0, ctx.cc, g)
            pull.for_y0.append(g)
        g_k1 = pull.take(0, g_k1c)
        g_y0 = pull.take_y0(g_y0c)
        return None, (g_y0 if need_y0 else None), (g_k1 if need_k else None), gc


class _RhsStagePullFn(torch.autograd.Function):
    """_RhsStageCarryFn in pull form: same forward (one ndcn_rhs_rk_f32 launch: K and the next stage input).

    The coefficients of a stage sum are ONE step size times the tableau row (c_j = dt beta_j: _step_coefficients), so what the scalar
    chain needs from this node is d/d dt = sum_j beta_j <g_u, k_j> = <g_u, u_next - y0> / dt - one ndcn_rk_dot_diff_f32 pass over three
    panels instead of one panel per term.  It is handed back on the LAST coefficient (the new K's: never zero in a tableau) as
    <g_u, u_next - y0> / c_last, the others get none: beta_last * that = the row's whole contribution to dt.  (NDCN_GRAD_ROW_DOT=0: one
    product per coefficient, as the push form has them.)"""

    @staticmethod
    def forward(ctx, n, pull, op, u, W, b, y0, *rest):
        ks, cs = rest[:n], rest[n:]
        A, no_graph, no_control = op
        idx, kk, cc = _active(ks, cs[:n])
        c_new = f32(float(cs[n]))
        S = _keep_s(op, 'combine', len(kk), u) if ctx.needs_input_grad[4] else None
        K, u_next = hip.rhs_rk(A, u, W, b, 'combine', y0, kk, cc + [c_new], no_graph=no_graph, no_control=no_control, s_out=S)
        ctx.S = S                                                     # (not an input or output of the node: kept on the context)
        ctx.n, ctx.idx, ctx.cc, ctx.op, ctx.has_b, ctx.pull = n, idx, cc + [c_new], op, b is not None, pull
        ctx.lazy = _LAZY.on
        ctx.row_dot = _row_dot_enabled() and float(c_new) != 0.0
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(u, W, K, u_next, y0, *kk, *cs)
        return (K, u_next, y0, W.view_as(W), (b.view_as(b) if b is not None else None)) + tuple(ks)

    @staticmethod
    def backward(ctx, g_K, g_u, g_y0c, g_Wc, g_bc, *g_kc):
        from ...autograd_ops import rhs_vjp
        n, idx, cc, pull = ctx.n, ctx.idx, ctx.cc, ctx.pull
        saved = ctx.saved_tensors
        u, W, K, u_next, y0s = saved[:5]
        saved = saved[2:]
        kk, cs = saved[3:3 + len(idx)], saved[3 + len(idx):]
        needs = ctx.needs_input_grad                                  # (n, pull, op, u, W, b, y0, k.., c..)
        need_u, need_w, need_b, need_y0 = needs[3], needs[4], ctx.has_b and needs[5], needs[6]
        need_k, need_c = needs[7:7 + n], needs[7 + n:]
        gc_all = [None] * (n + 1)
        if g_u is not None:
            g_u = g_u.contiguous()
            if ctx.row_dot and need_c[n]:
                d = hip.dot_diff(g_u, u_next, y0s, scale=1.0 / float(cc[len(idx)]), lazy=ctx.lazy)
                gc_all[n] = _grad_scalar(cs[n], d)
            elif any(need_c[j] for j in idx) or need_c[n]:
                res = hip.combine_bwd(g_u, list(kk) + [K], cc, [False] * (len(idx) + 1), need_dots=True, lazy=ctx.lazy)
                dots = res[1]
                for q, j in enumerate(idx):
                    if need_c[j]:
                        gc_all[j] = _grad_scalar(cs[j], dots[q])
                if need_c[n]:
                    gc_all[n] = _grad_scalar(cs[n], dots[len(idx)])
            for q, j in enumerate(idx):
                if need_k[j]:
                    pull.note(j, cc[q], g_u)
            pull.note(n, cc[len(idx)], g_u)
            if need_y0:
                pull.for_y0.append(g_u)
        g_K = pull.take(n, g_K)                                       # what this evaluation's K received: later stages, error, dense output
        gu_in = gW = gb = None
        if g_K is not None:
            A, no_graph, no_control = ctx.op
            gu_in, gW, gb = rhs_vjp(A, no_graph, no_control, u, W, K, g_K.contiguous(), need_u, need_w, need_b, S=ctx.S)
        ctx.S = None
        gW = _add_carried(gW, g_Wc) if need_w else None
        gb = _add_carried(gb, g_bc) if need_b else None
        gk_all = [g_kc[j] if need_k[j] else None for j in range(n)]    # handed through untouched: their producers pull
        return (None, None, None, gu_in, gW, gb, (g_y0c if need_y0 else None)) + tuple(gk_all) + tuple(gc_all)


class _RhsErrorCarryFn(torch.autograd.Function):
    """(K, ratio, y0', y1', k_1', ..) = (relu(W (A y1) + b), mean((sum_j c_j k_j + c_new K)^2 / tol^2), y0, y1, k_1, ..): the LAST
    evaluation of a dopri5 step with the error record in its epilogue (ndcn_rhs_rk_f32, mode error - the inference solver's
    launch), i.e. `_Rhs` followed by `_ErrorCarryFn` as one node.  Used where the inference path fuses the record too: panels
    beyond the ATen-order reductions' range (rk.hip: NDCN_ATEN_NORM_MAX), so both paths take identical accept / reject decisions."""

    @staticmethod
    def forward(ctx, n, op, rtol, atol, bad_out, W, b, y0, y1, *rest):
        ks, cs = rest[:n], rest[n:]                                   # n earlier stages, n + 1 coefficients (the new K last)
        A, no_graph, no_control = op
        idx, kk, cc = _active(ks, cs[:n])
        c_new = f32(float(cs[n]))
        S = _keep_s(op, 'error', len(kk), y1) if ctx.needs_input_grad[5] else None
        K, (s, bad) = hip.rhs_rk(A, y1, W, b, 'error', y0, kk, cc + [c_new], rtol=rtol, atol=atol, no_graph=no_graph, no_control=no_control,
                                 s_out=S)
        ctx.S = S
        bad_out.append(bad)
        ctx.n, ctx.idx, ctx.cc, ctx.op, ctx.has_b, ctx.tol = n, idx, cc + [c_new], op, b is not None, (rtol, atol)
        ctx.lazy = _LAZY.on
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(y0, y1, W, K, *kk, *cs)
        return (K, torch.tensor(f32(s / y0.numel()), dtype=torch.float32), y0, y1, W.view_as(W), (b.view_as(b) if b is not None else None)) + tuple(ks)

    @staticmethod
    def backward(ctx, g_K, g, g_y0c, g_y1c, g_Wc, g_bc, *g_kc):
        from ...autograd_ops import rhs_vjp
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        y0, y1, W, K = saved[:4]
        kk, cs = saved[4:4 + len(idx)], saved[4 + len(idx):]
        needs = ctx.needs_input_grad                                  # (n, op, rtol, atol, bad_out, W, b, y0, y1, k.., c..)
        need_w, need_b, need_y0, need_y1 = needs[5], ctx.has_b and needs[6], needs[7], needs[8]
        need_k, need_c = needs[9:9 + n], needs[9 + n:]
        gk_all = [g_kc[j] if need_k[j] else None for j in range(n)]
        gc_all = [None] * (n + 1)
        gy0, gy1 = (g_y0c if need_y0 else None), (g_y1c if need_y1 else None)
        if g is not None:
            g_r = float(g)
            want_dots = any(need_c[j] for j in idx) or need_c[n]
            gy0, gy1, gk, dots = hip.error_bwd(y0, y1, list(kk) + [K], cc, ctx.tol[0], ctx.tol[1], g_r, need_y0, need_y1,
                                               [need_k[j] for j in idx] + [True], need_dots=want_dots,
                                               accs=[g_kc[j] if need_k[j] else None for j in idx] + [g_K],
                                               acc_y0=g_y0c if need_y0 else None, acc_y1=g_y1c if need_y1 else None, lazy=ctx.lazy)
            for q, j in enumerate(idx):
                if need_k[j]:
                    gk_all[j] = gk[q]
                if need_c[j]:
                    gc_all[j] = dots[q] if ctx.lazy else _scalar_like(cs[j], g_r * dots[q])
            if need_c[n]:
                gc_all[n] = dots[len(idx)] if ctx.lazy else _scalar_like(cs[n], g_r * dots[len(idx)])
            g_K = gk[len(idx)]
        gW = gb = None
        if g_K is not None:
            A, no_graph, no_control = ctx.op
            gx, gW, gb = rhs_vjp(A, no_graph, no_control, y1, W, K, g_K.contiguous(), need_y1, need_w, need_b, S=ctx.S)
            if need_y1 and gx is not None:
                gy1 = gx if gy1 is None else hip.combine(gy1.contiguous(), [gx], [f32(1)])      # y1 is the evaluation's input AND the record's state
        gW = _add_carried(gW, g_Wc) if need_w else None
        gb = _add_carried(gb, g_bc) if need_b else None
        return (None, None, None, None, None, gW, gb, gy0, gy1) + tuple(gk_all) + tuple(gc_all)


class _ErrorCarryFn(torch.autograd.Function):
    """(ratio, y0', y1', k_1', ..)"""

    @staticmethod
    def forward(ctx, n, rtol, atol, bad_out, y0, y1, *rest):
        ks, cs = rest[:n], rest[n:]
        idx, kk, cc = _active(ks, cs)
        s, bad = hip.error(y0, y1, kk, cc, rtol, atol)
        bad_out.append(bad)
        ctx.n, ctx.idx, ctx.cc, ctx.tol = n, idx, cc, (rtol, atol)
        ctx.lazy = _LAZY.on
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(y0, y1, *kk, *cs)
        return (torch.tensor(f32(s / y0.numel()), dtype=torch.float32), y0, y1) + tuple(ks)

    @staticmethod
    def backward(ctx, g, g_y0c, g_y1c, *g_kc):
        n, idx, cc = ctx.n, ctx.idx, ctx.cc
        saved = ctx.saved_tensors
        y0, y1 = saved[:2]
        kk, cs = saved[2:2 + len(idx)], saved[2 + len(idx):]
        needs = ctx.needs_input_grad
        need_k, need_c = needs[6:6 + n], needs[6 + n:]
        gk_all = [g_kc[j] if need_k[j] else None for j in range(n)]
        gc_all = [None] * n
        gy0, gy1 = (g_y0c if needs[4] else None), (g_y1c if needs[5] else None)
        if g is not None:
            g_r = float(g)
            gy0, gy1, gk, dots = hip.error_bwd(y0, y1, kk, cc, ctx.tol[0], ctx.tol[1], g_r, needs[4], needs[5],
                                               [need_k[j] for j in idx], need_dots=any(need_c[j] for j in idx),
                                               accs=[g_kc[j] if need_k[j] else None for j in idx],
                                               acc_y0=g_y0c if needs[4] else None, acc_y1=g_y1c if needs[5] else None, lazy=ctx.lazy)
            for q, j in enumerate(idx):
                if need_k[j]:
                    gk_all[j] = gk[q]
                if need_c[j]:
                    gc_all[j] = dots[q] if ctx.lazy else _scalar_like(cs[j], g_r * dots[q])      # (lazy: g_r is folded in on the device)
        return (None, None, None, None, gy0, gy1) + tuple(gk_all) + tuple(gc_all)


class _DenseCarryFn(torch.autograd.Function):
    """(dense output, y0', y1', k_1', .., k_7')"""

    @staticmethod
    def forward(ctx, cache, a0, a1, *rest):
        kk, dt_, x_ = rest[:7], rest[7], rest[8]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a0, a1, *rest)
        return (_dense_value(cache, a0, a1, kk, dt_, x_), a0, a1) + tuple(kk)

    @staticmethod
    def backward(ctx, g, g_a0c, g_a1c, *g_kc):
        saved = ctx.saved_tensors
        a0, a1, kk, dt_, x_ = saved[0], saved[1], saved[2:9], saved[9], saved[10]
        needs = ctx.needs_input_grad
        need_k = list(needs[3:10])
        gk = [g_kc[j] if need_k[j] else None for j in range(7)]
        gy0, gy1 = (g_a0c if needs[1] else None), (g_a1c if needs[2] else None)
        g_dt = g_x = None
        if g is not None:
            gy0, gy1, gk, d_x, d_dt = hip.interp_bwd(g.contiguous(), a0, a1, list(kk), f32(float(dt_)), f32(float(x_)), needs[1], needs[2],
                                                     need_k, need_dots=needs[10] or needs[11],
                                                     accs=[g_kc[j] if need_k[j] else None for j in range(7)],
                                                     acc_y0=g_a0c if needs[1] else None, acc_y1=g_a1c if needs[2] else None)
            g_dt = _scalar_like(dt_, d_dt) if needs[10] else None
            g_x = _scalar_like(x_, d_x) if needs[11] else None
        return (None, gy0, gy1) + tuple(gk) + (g_dt, g_x)


class _DenseMultiCarryFn(torch.autograd.Function):
    """(o_1, .., o_nt, y0', y1', k_1', .., k_7'): nt <= 7 ticks of ONE accepted step in one pass (forward: fit + evaluate per
    tick, the bits of the single-tick form; backward: the step's nine panels and their received gradients are read once)."""

    @staticmethod
    def forward(ctx, nt, a0, a1, *rest):
        kk, dt_, xs = rest[:7], rest[7], rest[8:8 + nt]
        ctx.nt = nt
        ctx.lazy = _LAZY.on
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a0, a1, *rest)
        dt32 = f32(float(dt_))
        cmid = [f32(dt32 * f32(c)) for c in core.DP_C_MID]
        xpows = []
        for x_ in xs:
            xv = f32(float(x_))
            x2 = f32(xv * xv)
            x3 = f32(x2 * xv)
            xpows.append((f32(x3 * xv), x3, x2, xv, f32(1)))
        outs = hip.interp_direct_multi(a0, a1, list(kk), cmid, dt32, xpows)
        return tuple(outs) + (a0, a1) + tuple(kk)

    @staticmethod
    def backward(ctx, *gs):
        nt = ctx.nt
        saved = ctx.saved_tensors
        a0, a1, kk, dt_, xs = saved[0], saved[1], saved[2:9], saved[9], saved[10:10 + nt]
        g_out, g_a0c, g_a1c, g_kc = gs[:nt], gs[nt], gs[nt + 1], gs[nt + 2:nt + 9]
        needs = ctx.needs_input_grad
        need_k = list(needs[3:10])
        gk = [g_kc[j] if need_k[j] else None for j in range(7)]
        gy0, gy1 = (g_a0c if needs[1] else None), (g_a1c if needs[2] else None)
        g_dt, g_x = None, [None] * nt
        live = [t for t in range(nt) if g_out[t] is not None]
        if live:
            gy0, gy1, gk, dxs, d_dt = hip.interp_bwd_multi([g_out[t].contiguous() for t in live], a0, a1, list(kk), f32(float(dt_)),
                                                           [f32(float(xs[t])) for t in live], needs[1], needs[2], need_k,
                                                           accs=[g_kc[j] if need_k[j] else None for j in range(7)],
                                                           acc_y0=g_a0c if needs[1] else None, acc_y1=g_a1c if needs[2] else None, lazy=ctx.lazy)
            g_dt = _grad_scalar(dt_, d_dt) if needs[10] else None
            for q, t in enumerate(live):
                if needs[11 + t]:
                    g_x[t] = _grad_scalar(xs[t], dxs[q])
        return (None, gy0, gy1) + tuple(gk) + (g_dt,) + tuple(g_x)


# ---- the solver, scalar chain in torch exactly as the reference keeps it -----------------------------------------

def _initial_step(func, targ, t0, y0, order, rtol, atol, f0, bad_out):
    """misc.py:84-143; returns a float32 0-d tensor with autograd history through the three norms.  Every state tensor has four
    consumers in here and every f0 three: each gets an alias of its own (_FanOut: gradients added in a fixed order)."""
    ya = [_fan(y, 4) for y in y0]
    fa = [_fan(f, 3) for f in f0]
    d0 = [_rms(y[0], None, y[0], rtol, atol, bad_out) for y in ya]
    d1 = [_rms(f[0], None, y[1], rtol, atol) for f, y in zip(fa, ya)]
    if max(d0).item() < 1e-5 or max(d1).item() < 1e-5:
        h0 = torch.tensor(1e-6, dtype=torch.float32)
    else:
        h0 = 0.01 * max(a / b for a, b in zip(d0, d1))
    hs = _fan(h0, 2 + len(y0)) if h0.requires_grad else (h0,) * (2 + len(y0))     # (combine per state tensor, d2, the result)
    y1 = tuple(_combine(y[2], [f[1]], [hs[2 + j]]) for j, (y, f) in enumerate(zip(ya, fa)))
    f1 = func(targ(f32(t0) + f32(h0.item())), y1)
    d2 = [_rms(b, f[2], y[3], rtol, atol) / hs[0] for b, f, y in zip(f1, fa, ya)]
    if max(d1).item() <= 1e-15 and max(d2).item() <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=torch.float32), hs[1] * 1e-3)
    else:
        h1 = (0.01 / max(d1 + d2)) ** (1. / float(order + 1))
    return torch.min(100 * hs[1], h1)


class _Sqrt32(torch.autograd.Function):
    """sqrt of a float32 0-d host tensor, CORRECTLY ROUNDED (numpy / libm sqrtf - what the inference solvers and the reference's
    fixtures have).  torch.sqrt on a CPU with AVX-512 returns values one ulp off (PyTorch 2.10 + AVX512 dispatch, measured:
    tools/micro/sqrt_probe.py - sqrt(5.2355666e-08f) = 2.28813617e-04 instead of 2.28813602e-04), which moved dt_next of the
    training path by 1e-8 against the inference path's on such hosts.  Backward as torch's: g / (2 sqrt(x))."""

    @staticmethod
    def forward(ctx, x):
        out = torch.tensor(np.sqrt(f32(float(x))), dtype=torch.float32)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        return g / (2 * out)


def integrate_dopri5_grad(*args, **kwargs):
    before = (_LAZY.on, _LAZY.keep_s)
    try:
        return _integrate_dopri5_grad(*args, **kwargs)
    finally:
        _LAZY.on, _LAZY.keep_s = before


def _integrate_dopri5_grad(func, y0, t, rtol, atol, autonomous=False, step_log=None, odefunc=None, **options):
    core.assert_increasing(t)
    dtype = y0[0].dtype
    targ = core.TimeArg(y0[0], autonomous)
    tt = core.host_grid(t).to(torch.float64)
    opt = core.dopri5_options(options, len(y0))                    # same validation / warnings as the inference path
    max_steps = opt['max_num_steps']
    safety = torch.tensor(opt['safety'], dtype=torch.float64)
    ifactor, dfactor = opt['ifactor'], opt['dfactor']
    rtols, atols = core.per_state_tolerance(rtol, len(y0)), core.per_state_tolerance(atol, len(y0))
    rtol, atol = rtols[0], atols[0]                                # dopri5.py:80: the initial step uses the first pair
    bad = []
    carry = _carry()
    # deferred scalar gradients pay where a read-back stalls real GPU work; on the reference's own sizes (400 x 20) the extra
    # autograd node per coefficient costs more than the stall (README-sized dopri5 step 11 -> 19 ms, tools/micro/train_ab.py)
    lazy = carry and _lazy() and y0[0].numel() >= int(os.environ.get('NDCN_GRAD_LAZY_MIN', 1 << 20))
    # a plain ODEFunc on one state tensor (odeint checked): its evaluations carry the next stage input in their epilogue
    fused = None
    # (reference-sized states - 400 x 20 - are host-bound: the one-node-per-evaluation form costs them 35 %, 14.1 against 10.4 ms
    # per README-sized dopri5 step, profiles/r04i_train_ab.txt; they keep one node per operation)
    if carry and odefunc is not None and len(y0) == 1 and os.environ.get('NDCN_GRAD_FUSED_STAGE', '1') != '0' and \
            y0[0].numel() >= int(os.environ.get('NDCN_GRAD_FUSED_STAGE_MIN', 1 << 16)):
        from ...csr import as_csr
        fused = ((None if odefunc.no_graph else as_csr(odefunc.A), bool(odefunc.no_graph), bool(odefunc.no_control)),
                 odefunc.wt.weight, odefunc.wt.bias)
    # the error record rides in the last evaluation's epilogue where the inference solver puts it there too (rk.hip: panels beyond
    # the range of the ATen-order reductions), so that both paths see the same ratio
    fuse_err = fused is not None and y0[0].numel() > int(os.environ.get('NDCN_ATEN_NORM_MAX', 1 << 18)) and \
        os.environ.get('NDCN_GRAD_FUSED_ERROR', '1') != '0'
    _LAZY.on = lazy
    pull_on = fused is not None and _pull_enabled()
    _LAZY.keep_s = fused is not None and _keep_s_enabled(y0[0])
    multi_tick = os.environ.get('NDCN_GRAD_MULTI_TICK', '1') != '0'
    # ---- the first evaluation and the initial step (dopri5.py:76-83).  y0 has four consumers - this evaluation, the initial-step
    # selection, the first stage chain, the trajectory's first tick: one alias each (_FanOut: a fixed summation order for its
    # gradient); in the fused path every evaluation goes through a carry node that hands W and b on (a chain instead of a dozen
    # contributions to the leaf in whatever order the engine's two threads produce them)
    first = opt['first_step'] is None
    ys = [_fan(y, 4 if first else 3) for y in y0] if carry else [(y,) * 4 for y in y0]
    chain = list(fused[1:]) if fused is not None else None             # [W, b] as handed on from evaluation to evaluation

    def evaluate(tval, yy):
        if fused is None:
            return func(targ(tval), yy)
        K_, chain[0], chain[1] = _RhsCarryFn.apply(fused[0], yy[0], chain[0], chain[1])
        return (K_,)

    f_cur = evaluate(f32(tt[0].item()), tuple(y[0] for y in ys))
    nfe = 2
    if first:
        dt = _initial_step(lambda targ_t, yy: evaluate(targ_t, yy), (lambda v: v), tt[0].item(), tuple(y[3] for y in ys), 4, rtol, atol, f_cur, bad).to(torch.float64)
    else:
        dt = torch.tensor(0.01, dtype=torch.float64)
        bad.append(0)
    pending_bad = bad[0] if bad else 0
    y_cur = tuple(y[1] for y in ys)
    t_lo = t_hi = tt[0]
    stage = None
    sol = [tuple(y[2] for y in ys)]
    i = 0
    while i + 1 < len(tt):
        i += 1
        nxt = tt[i]
        n_steps = 0
        while nxt.item() > t_hi.item():
            assert n_steps < max_steps, 'max_num_steps exceeded ({}>={})'.format(n_steps, max_steps)
            t0 = t_hi
            assert (t0 + dt).item() > t0.item(), 'underflow in dt {}'.format(dt.item())
            assert pending_bad == 0, 'non-finite values in state `y`: {} elements'.format(int(pending_bad))
            t0s, dts = t0.to(dtype), dt.to(dtype)                      # rk_common.py:45-46
            k = [[f] for f in f_cur]
            yi = y_cur
            yc = list(y_cur)                                           # carry mode: the step's y0 as handed on from op to op
            # (one coefficient tensor per consuming operation; created where they are used, i.e. behind queued GPU work - not
            # in front of the step's first launch, where 50 tiny host operations would sit between the accept decision's
            # read-back and the next kernel)
            coef = (lambda v: _Await.apply(v)) if lazy else (lambda v: v)
            beta_dt, cerr_dt = _step_coefficients(dts)
            fused_bads, fused_ratio = [], None
            if fused is not None:
                # stage input 1 by a combine launch; evaluations 2 .. 6 form the next stage input themselves; evaluation 7 (at
                # y1, the next step's k1) is the plain right-hand side
                op_ = fused[0]
                pull_ = _StepPull() if pull_on else None
                if pull_ is not None:
                    outs_ = _StagePullFn.apply(pull_, yc[0], k[0][0], coef(beta_dt[0][0]))
                else:
                    outs_ = _StageCarryFn.apply(1, yc[0], k[0][0], coef(beta_dt[0][0]))
                u_, yc[0], k[0] = outs_[0], outs_[1], list(outs_[2:])
                for st_ in range(6):
                    if st_ < 5:
                        if pull_ is not None:
                            outs_ = _RhsStagePullFn.apply(len(k[0]), pull_, op_, u_, chain[0], chain[1], yc[0], *k[0],
                                                          *[coef(v) for v in beta_dt[st_ + 1]])
                        else:
                            outs_ = _RhsStageCarryFn.apply(len(k[0]), op_, u_, chain[0], chain[1], yc[0], *k[0],
                                                           *[coef(v) for v in beta_dt[st_ + 1]])
                        u_, yc[0], chain[0], chain[1], k[0] = outs_[1], outs_[2], outs_[3], outs_[4], list(outs_[5:]) + [outs_[0]]
                    elif fuse_err:
                        yi = (u_,)                                    # evaluation 7 with the error record in its epilogue
                        outs_ = _RhsErrorCarryFn.apply(len(k[0]), op_, rtols[0], atols[0], fused_bads, chain[0], chain[1], yc[0], u_, *k[0],
                                                       *[coef(v) for v in cerr_dt])
                        fused_ratio, yc[0], yi, chain[0], chain[1], k[0] = outs_[1], outs_[2], (outs_[3],), outs_[4], outs_[5], list(outs_[6:]) + [outs_[0]]
                    else:
                        yi = (u_,)
                        k[0].append(evaluate((t0s + core.DP_ALPHA[5] * dts).item(), yi)[0])
                    nfe += 1
            for (a_i, _), b_i in (() if fused is not None else zip(zip(core.DP_ALPHA, core.DP_BETA), beta_dt)):
                ti = t0s + a_i * dts
                if carry:
                    us = []
                    for s_, (y_, k_) in enumerate(zip(yc, k)):
                        outs_ = _StageCarryFn.apply(len(k_), y_, *k_, *[coef(v) for v in b_i])
                        us.append(outs_[0])
                        yc[s_], k[s_] = outs_[1], list(outs_[2:])
                    yi = tuple(us)
                else:
                    yi = tuple(_combine(y_, k_, list(b_i)) for y_, k_ in zip(y_cur, k))
                for k_, f_ in zip(k, func(targ(ti.item()), yi)):
                    k_.append(f_)
                nfe += 1
            y1 = yi
            bads = []
            if fused is not None and fuse_err:
                ratios, bads = [fused_ratio], fused_bads
            elif carry:
                ratios, y1c = [], []
                for s_, (a_, b_, k_, rt_, at_) in enumerate(zip(yc, y1, k, rtols, atols)):
                    outs_ = _ErrorCarryFn.apply(len(k_), rt_, at_, bads, a_, b_, *k_, *[coef(v) for v in cerr_dt])
                    ratios.append(outs_[0])
                    yc[s_], k[s_] = outs_[1], list(outs_[3:])
                    y1c.append(outs_[2])
                y1 = tuple(y1c)
            else:
                ratios = [_error_ratio(a_, b_, k_, list(cerr_dt), rt_, at_, bads)
                          for a_, b_, k_, rt_, at_ in zip(y_cur, y1, k, rtols, atols)]
            f1 = tuple(k_[-1] for k_ in k)
            accept = bool((torch.stack([r.detach() for r in ratios]) <= 1).all())      # dopri5.py:109
            worst = max(ratios)
            # misc.py:160-170
            if worst.item() == 0:
                dt_next = dt * ifactor
            else:
                dfac = 1.0 if worst.item() < 1 else dfactor
                er = _Sqrt32.apply(worst).to(torch.float64)
                expo = torch.tensor(1 / 5).to(torch.float64)
                factor = torch.max(torch.tensor(1 / ifactor, dtype=torch.float64),
                                   torch.min(er ** expo / safety, torch.tensor(1 / dfac, dtype=torch.float64)))
                dt_next = dt / factor
            if step_log is not None:
                step_log.append((t0.item(), dt.item(), 1.0 if accept else 0.0, worst.item(), dt_next.item()))
            if accept:
                stage = (tuple(yc) if carry else y_cur, y1, k, dts, {})
                y_cur, f_cur = y1, f1
                t_lo, t_hi = t0, t0 + dt
                pending_bad = sum(bads)
            else:
                t_lo = t_hi = t0
                if carry:                                              # the retry reads what the rejected attempt handed on
                    y_cur, f_cur = tuple(yc), tuple(k_[0] for k_ in k)
            dt = dt_next
            n_steps += 1
        # interp.py:51-65
        a0, a1, at = t_lo.to(dtype), t_hi.to(dtype), nxt.to(dtype)
        assert (a0.item() <= at.item()) and (at.item() <= a1.item()), \
            'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(a0, at, a1)
        x = ((at - a0) / (a1 - a0)).to(dtype)
        s_y0, s_y1, s_k, s_dts, caches = stage
        outs = []
        if carry and multi_tick:
            # every tick this accepted step covers (<= 7 per operation) in ONE pass over the step's panels
            xs = [x]
            while i + 1 < len(tt) and len(xs) < 7 and not (tt[i + 1].item() > t_hi.item()):
                i += 1
                at = tt[i].to(dtype)
                assert (a0.item() <= at.item()) and (at.item() <= a1.item()), \
                    'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(a0, at, a1)
                xs.append(((at - a0) / (a1 - a0)).to(dtype))
            nt_ = len(xs)
            per_tick = [[] for _ in range(nt_)]
            n_y0, n_y1, n_k = [], [], []
            for j, (p0, p1, kk) in enumerate(zip(s_y0, s_y1, s_k)):
                o_ = _DenseMultiCarryFn.apply(nt_, p0, p1, *kk, *([_Await.apply(s_dts)] + [_Await.apply(x_) for x_ in xs] if lazy else [s_dts] + xs))
                for q in range(nt_):
                    per_tick[q].append(o_[q])
                n_y0.append(o_[nt_]); n_y1.append(o_[nt_ + 1]); n_k.append(list(o_[nt_ + 2:]))
            stage = (tuple(n_y0), tuple(n_y1), n_k, s_dts, caches)
            y_cur, f_cur = tuple(n_y1), tuple(k_[-1] for k_ in n_k)
            for q in range(nt_ - 1):
                sol.append(tuple(per_tick[q]))
            outs = per_tick[-1]
        elif carry:
            n_y0, n_y1, n_k = [], [], []
            for j, (p0, p1, kk) in enumerate(zip(s_y0, s_y1, s_k)):
                o_ = _DenseCarryFn.apply(caches.setdefault(j, {}), p0, p1, *kk, s_dts, x)
                outs.append(o_[0])
                n_y0.append(o_[1]); n_y1.append(o_[2]); n_k.append(list(o_[3:]))
            # the next tick (or the next step, whose y0 / k1 are this step's y1 / k7) continues from the handed-on tensors
            stage = (tuple(n_y0), tuple(n_y1), n_k, s_dts, caches)
            y_cur, f_cur = tuple(n_y1), tuple(k_[-1] for k_ in n_k)
        else:
            for j, (p0, p1, kk) in enumerate(zip(s_y0, s_y1, s_k)):
                outs.append(_dense_output(p0, p1, kk, s_dts, x, caches.setdefault(j, {})))
        sol.append(tuple(outs))
    if step_log is not None:
        step_log.append(('nfe', nfe))
    return sol
