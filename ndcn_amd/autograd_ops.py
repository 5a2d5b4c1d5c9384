"""Differentiable wrappers of the HIP ops: backward of the path.

The reference trains by plain autograd THROUGH every solver op (heat_dynamics.py:333; no adjoint).  Here
each forward is the same HIP kernel the inference path uses; the backward of the two heavy ops runs on
HIP kernels too (g_X = A^T g through the SpMM on the transposed CSR, g_S = g W through the MFMA Linear);
the weight gradient g^T S as a split-row fp32-MFMA kernel with the ReLU mask and the bias gradient fused in
(csrc/linear_bwd.hip)), and the per-term scalings of the Runge-Kutta algebra are one streaming HIP kernel each.

dopri5: the reference differentiates through its step-size controller too; that chain lives in
ndcn_amd/torchdiffeq/_impl/autograd_path.py.  The wrappers below serve the right-hand side and the fixed-grid
solvers, whose step sizes are data-independent.
"""
import numpy as np
import torch

from .csr import as_csr
from .ops import hip


class _Spmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, A):
        ctx.A = A
        return hip.spmm(A, X.detach())

    @staticmethod
    def backward(ctx, g):
        return hip.spmm(ctx.A.transpose(), g.contiguous()), None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, W, b):
        ctx.save_for_backward(S, W)
        ctx.has_b = b is not None
        return hip.linear(S.detach(), W.detach(), None if b is None else b.detach())

    @staticmethod
    def backward(ctx, g):
        S, W = ctx.saved_tensors
        return hip.linear_bwd(g.contiguous(), W.detach(), S=S.detach(), need_gS=ctx.needs_input_grad[0],
                              need_gW=ctx.needs_input_grad[1], need_gb=ctx.has_b and ctx.needs_input_grad[2])


class _Rhs(torch.autograd.Function):
    """relu(W (A X) + b) with the fused forward; S = A X is recomputed in the backward instead of stored."""

    @staticmethod
    def forward(ctx, X, W, b, A, no_graph, no_control):
        # (W itself, not a detached alias: the packed-weight cache of ops.rhs is keyed on the tensor object and its version
        # counter - every evaluation of a solve then reuses the packed image instead of rebuilding it: two launches per RHS)
        Y = hip.rhs(A, X.detach(), W, None if b is None else b.detach(), no_graph=no_graph, no_control=no_control)
        ctx.A, ctx.no_graph, ctx.no_control, ctx.has_b = A, no_graph, no_control, b is not None
        ctx.save_for_backward(X, W, Y)
        return Y

    @staticmethod
    def backward(ctx, g):
        X, W, Y = ctx.saved_tensors
        gX, gW, gb = rhs_vjp(ctx.A, ctx.no_graph, ctx.no_control, X, W, Y, g.contiguous(), ctx.needs_input_grad[0],
                             ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2])
        return gX, gW, gb, None, None, None


def rhs_vjp(A, no_graph, no_control, X, W, Y, g, need_x, need_w, need_b, S=None):
    """(g_X, g_W, g_b) of Y = relu(W (A X) + b) for the upstream gradient g - the closed form every differentiable wrapper of the
    right-hand side shares: ReLU mask fused into the operand loads of the two GEMMs, g_W = gZ^T S split over row chunks (S = A X
    recomputed instead of stored, unless the caller kept the one the forward launch wrote), g_b with it, g_X = A^T g_S through the
    SpMM on the transposed CSR."""
    gW = gb = None
    if not no_control:
        if not need_w:
            S = None
        elif S is None:
            S = X.detach() if no_graph else hip.spmm(A, X.detach())
        # (W itself, not a detached alias: linear_bwd keeps the packed planes of W^T per weight tensor object and version)
        gS, gW, gb = hip.linear_bwd(g, W, S=S, Y=Y, need_gS=need_x, need_gW=need_w, need_gb=need_b)
    else:
        gS = hip.relu_bwd(g, Y) if need_x else None
    gX = None
    if need_x:
        gX = gS if no_graph else hip.spmm(A.transpose(), gS.contiguous())
    return gX, gW, gb


def spmm(A, x):
    return _Spmm.apply(x, as_csr(A))


def linear(x, W, b):
    return _Linear.apply(x, W, b)


def rhs(A, x, W, b, no_graph, no_control):
    return _Rhs.apply(x, W, b, None if no_graph else as_csr(A), no_graph, no_control)


# ---------------------------------------------------------------------------------------------------
# Runge-Kutta algebra: linear maps whose VJP is a per-input scalar times the incoming gradient
# ---------------------------------------------------------------------------------------------------

class _LinMap(torch.autograd.Function):
    """forward = a HIP kernel given as closure; out = sum_i w_i * in_i with host-side scalar weights w."""

    @staticmethod
    def forward(ctx, fwd, weights, *inputs):
        ctx.weights = weights
        return fwd(*[t.detach() for t in inputs])

    @staticmethod
    def backward(ctx, g):
        grads = []
        for need, w in zip(ctx.needs_input_grad[2:], ctx.weights):
            grads.append(hip.scale(g, w) if (need and w != 0) else (torch.zeros_like(g) if need else None))
        return (None, None) + tuple(grads)


# fixed_stage op -> weights of (y, k1, k2, k3, k4) as functions of dt (the formulas in include/ndcn_hip.h)
_STAGE_W = {
    0: lambda dt: (1, dt),
    1: lambda dt: (1, dt / 2),
    2: lambda dt: (1, dt / 3),
    3: lambda dt: (1, -dt / 3, dt),
    4: lambda dt: (1, dt, -dt, dt),
    5: lambda dt: (1, dt / 8, 3 * dt / 8, 3 * dt / 8, dt / 8),
}


class AutogradOps:
    """Same interface as ndcn_amd.ops.HipOps; every panel op is differentiable."""

    name = 'hip+autograd'

    @staticmethod
    def combine(y0, ks, cs):
        w = (1.0,) + tuple(float(c) for c in cs)
        return _LinMap.apply(lambda y, *k: hip.combine(y, list(k), cs), w, y0, *ks)

    @staticmethod
    def scale(x, w):
        return _LinMap.apply(lambda xx: hip.scale(xx, w), (float(w),), x)

    @staticmethod
    def fixed_stage(op, y, k1, k2=None, k3=None, k4=None, dt=0.0, out=None):
        w = _STAGE_W[op](float(dt))
        ins = [t for t in (y, k1, k2, k3, k4) if t is not None][:len(w)]
        return _LinMap.apply(lambda yy, *k: hip.fixed_stage(op, yy, *k, dt=dt), w, *ins)

    @staticmethod
    def error(y0, y1, ks, cs, rtol, atol):
        return hip.error(y0.detach(), y1.detach(), [k.detach() for k in ks], cs, rtol, atol)

    @staticmethod
    def scaled_sumsq(a, b, y, rtol, atol):
        return hip.scaled_sumsq(a.detach(), None if b is None else b.detach(), y.detach(), rtol, atol)

    @staticmethod
    def interp_fit(y0, y1, ks, cmid, dt):
        with torch.no_grad():
            abcd = hip.interp_fit(y0.detach(), y1.detach(), [k.detach() for k in ks], cmid, dt)
        return {'abcd': abcd, 'y0': y0, 'y1': y1, 'ks': list(ks), 'cmid': [float(c) for c in cmid], 'dt': float(dt)}

    @staticmethod
    def interp_eval(fit, e, xpow, out=None):
        x4, x3, x2, x1, _ = [float(v) for v in xpow]
        dt = fit['dt']
        w_f0 = dt * (-2 * x4 + 5 * x3 - 4 * x2 + x1)
        w_f1 = dt * (2 * x4 - 3 * x3 + x2)
        w_y0 = -8 * x4 + 18 * x3 - 11 * x2 + 1
        w_y1 = -8 * x4 + 14 * x3 - 5 * x2
        w_ym = 16 * x4 - 32 * x3 + 16 * x2
        wk = [w_ym * c for c in fit['cmid']]
        wk[0] += w_f0
        wk[-1] += w_f1
        weights = (w_y0 + w_ym, w_y1) + tuple(wk)
        abcd = fit['abcd']
        return _LinMap.apply(lambda y0, y1, *k: hip.interp_eval(abcd, y0, xpow), weights, fit['y0'], fit['y1'], *fit['ks'])

    # row-local pieces used by sharded / custom funcs
    spmm = staticmethod(lambda A, X, X_halo=None, alpha=1.0, relu=False, out=None: spmm(A, X))
    linear = staticmethod(lambda S, W, b=None, relu=False: linear(S, W, b))


autograd_ops = AutogradOps()
