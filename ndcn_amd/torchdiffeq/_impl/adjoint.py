"""`odeint_adjoint` - O(1)-memory backward by integrating the adjoint system backwards in time
(reference: torchdiffeq/_impl/adjoint.py:7-133).

Forward: this package's `odeint` without autograd history (device-resident HIP solver when `func` is ODEFunc).
Backward: for every tick interval [t_i, t_{i-1}], the augmented state (y, a_y, a_t, a_theta) is integrated
backwards with the same solver settings; its right-hand side evaluates func once and takes one vector-Jacobian
product of it (HIP SpMM with the transposed operator / MFMA Linear through autograd_ops).  Only the saved
trajectory (T x N x H) is kept between the two passes.
"""
import torch
import torch.nn as nn

from . import core
from .odeint import odeint


def _flat(tensors, like=None):
    if like is None:
        parts = [p.contiguous().view(-1) for p in tensors]
    else:
        parts = [p.contiguous().view(-1) if p is not None else torch.zeros_like(q).view(-1) for p, q in zip(tensors, like)]
    return torch.cat(parts) if parts else torch.tensor([])


class _TupleFunc(nn.Module):
    def __init__(self, base_func):
        super().__init__()
        self.base_func = base_func
        self.ndcn_autonomous = bool(getattr(base_func, 'ndcn_autonomous', False))

    def forward(self, t, y):
        return (self.base_func(t, y[0]),)


def _native_odefunc(func, n_state, ans):
    """This package's ODEFunc behind the tuple wrapper, when the closed-form adjoint right-hand side applies: one N x H
    device panel as state, the deterministic ODEFunc (dropout inactive), a square operator."""
    import os
    from ...neural_dynamics import ODEFunc
    base = getattr(func, 'base_func', None)
    if os.environ.get('NDCN_ADJOINT_NATIVE', '1') == '0' or n_state != 1 or type(base) is not ODEFunc:
        return None
    y = ans[0]
    if y.dim() != 3 or y.shape[2] != base.hidden_size or not y.is_cuda or (base.training and base.dropout > 0):
        return None
    return base


class _AdjointMethod(torch.autograd.Function):

    @staticmethod
    def forward(ctx, func, user_func, n_state, t, flat_params, rtol, atol, method, options, *y0):
        ctx.func, ctx.rtol, ctx.atol, ctx.method, ctx.options, ctx.n_state = func, rtol, atol, method, options, n_state
        with torch.no_grad():
            if user_func is not None:                      # single-tensor state: let the fast path see ODEFunc itself
                ans = (odeint(user_func, y0[0], t, rtol=rtol, atol=atol, method=method, options=options),)
            else:
                ans = odeint(func, tuple(y0), t, rtol=rtol, atol=atol, method=method, options=options)
        ctx.save_for_backward(t, flat_params, *ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        t, flat_params, *ans = ctx.saved_tensors
        func, rtol, atol, method, options = ctx.func, ctx.rtol, ctx.atol, ctx.method, ctx.options
        n = ctx.n_state
        f_params = tuple(func.parameters())

        def augmented(tt, y_aug):
            # d/dt (y, a_y, a_t, a_theta) = (f, -a_y^T df/dy, -a_y^T df/dt, -a_y^T df/dtheta)   adjoint.py:34-59
            y, adj_y = y_aug[:n], y_aug[n:2 * n]
            with torch.enable_grad():
                tt = tt.to(y[0].device).detach().requires_grad_(True)
                y = tuple(v.detach().requires_grad_(True) for v in y)
                f_eval = func(tt, y)
                vjp_t, *rest = torch.autograd.grad(f_eval, (tt,) + y + f_params, tuple(-a for a in adj_y),
                                                   allow_unused=True, retain_graph=True)
            vjp_y, vjp_p = rest[:n], rest[n:]
            vjp_t = torch.zeros_like(tt) if vjp_t is None else vjp_t
            vjp_y = tuple(torch.zeros_like(v) if g is None else g for g, v in zip(vjp_y, y))
            vjp_p = _flat(vjp_p, f_params)
            if len(f_params) == 0:
                vjp_p = torch.tensor(0.).to(vjp_y[0])
            return (*[f.detach() for f in f_eval], *vjp_y, vjp_t, vjp_p)

        native = _native_odefunc(func, n, ans)
        if native is not None:
            # HIP-native right-hand side (csrc/adjoint.hip): func_eval and the three vector-Jacobian products as closed forms
            # of ODEFunc = relu(W (A y) + b) - no torch graph, no zeros_like, no torch reductions on the path
            from ...ops import hip
            f0 = native
            zero_t = torch.zeros((), dtype=t.dtype, device=ans[0].device)
            no_params = torch.zeros((), dtype=ans[0].dtype, device=ans[0].device)

            def augmented(tt, y_aug):                                                    # noqa: F811
                K, vjp_y, vW, vb = hip.adjoint_rhs(f0.A, y_aug[0], y_aug[1], f0.wt.weight, f0.wt.bias, f0.no_graph, f0.no_control)
                if vW is None:                                   # no_control: `wt` sits in the parameter list, unused -> zeros
                    vjp_p = torch.zeros_like(flat_params)
                else:
                    vjp_p = torch.cat([vW.view(-1), vb]) if f0.wt.bias is not None else vW.view(-1)
                return (K, vjp_y, zero_t, vjp_p if len(f_params) else no_params)

        fused = None
        if native is not None and method in (None, 'dopri5'):
            # the reverse pass on the fused launches of the inference path (adjoint_fused.py): stage algebra in the epilogues of the
            # forward launch and of the transposed launch; the closure above serves the initial step of every interval only
            from . import adjoint_fused
            # integrate_interval runs in tau = -t from -t[i] UP to -t[i-1]: an increasing grid only.  A decreasing grid (the forward
            # pass went through odeint's sign flip, misc.py:184-187) keeps the generic odeint(augmented, ...) branch below
            th = core.host_grid(t)
            if bool((th[1:] > th[:-1]).all()) and adjoint_fused.applicable(native, ans[0]):
                fused = adjoint_fused
                fused_w = adjoint_fused._Weights(native)
                native_rhs = augmented

                def reversed_rhs(tau, y_aug):                    # misc.py:184-187: func(-t, y) negated
                    return tuple(-v for v in native_rhs(-tau, y_aug))

        fused_fixed = None
        if native is not None and method in ('euler', 'midpoint', 'rk4') and not (options or {}):
            # fixed grids: one step per interval on the same launches, the method's stage algebra in their epilogues (any width)
            from . import adjoint_fused
            th = core.host_grid(t)
            if bool((th[1:] > th[:-1]).all()) and adjoint_fused.applicable_fixed(native, ans[0]):
                fused_fixed = adjoint_fused
                fused_w = adjoint_fused._Weights(native)
                t_host = th.tolist()
        T = ans[0].shape[0]
        # panels beyond the ATen-order bound are reduced in parallel anyway: let the 65 792-element parameter-gradient vector riding
        # next to them follow (0.6 ms serial per reduction otherwise; include/ndcn_hip.h: ndcn_set_aten_norm_max)
        restore = None
        if fused is not None and ans[0][0].numel() > (1 << 18):
            from ... import _lib
            restore = (_lib.load(), _lib.load().ndcn_set_aten_norm_max(0))
        try:
            with torch.no_grad():
                adj_y = tuple(g[-1] for g in grad_output)
                adj_params = torch.zeros_like(flat_params)
                adj_time = torch.tensor(0.).to(t)
                time_vjps = []
                for i in range(T - 1, 0, -1):
                    ans_i = tuple(a[i] for a in ans)
                    grad_i = tuple(g[i] for g in grad_output)
                    f_i = func(t[i], ans_i)
                    # effect of moving the measurement time                               adjoint.py:72-77
                    if native is not None:                           # <f, g> by the library's fixed-order reduction (fp64 partials)
                        _, dots = hip.combine_bwd(grad_i[0].contiguous(), [f_i[0]], [1.0], [False], need_dots=True)
                        dLd_t = torch.tensor([dots[0]], dtype=t.dtype, device=adj_y[0].device)
                    else:
                        dLd_t = sum(torch.dot(f.reshape(-1), g.reshape(-1)).view(1) for f, g in zip(f_i, grad_i))
                    adj_time = adj_time - dLd_t
                    time_vjps.append(dLd_t)
                    if adj_params.numel() == 0:
                        adj_params = torch.tensor(0.).to(adj_y[0])
                    if fused is not None:
                        _, a_lo, adj_time, adj_params = fused.integrate_interval(
                            hip, native, fused_w, ans_i[0].contiguous(), adj_y[0].contiguous(), adj_time, adj_params, t[i], t[i - 1],
                            rtol, atol, options, reversed_rhs, step_log=getattr(native, 'ndcn_adjoint_step_log', None))
                        adj_y = (a_lo,)
                        aug0 = aug = None
                    elif fused_fixed is not None:
                        _, a_lo, adj_time, adj_params = fused_fixed.integrate_interval_fixed(
                            hip, native, fused_w, ans_i[0].contiguous(), adj_y[0].contiguous(), adj_time, adj_params, t_host[i], t_host[i - 1],
                            method, None, step_log=getattr(native, 'ndcn_adjoint_step_log', None))
                        adj_y = (a_lo,)
                        aug0 = aug = None
                    else:
                        aug0 = (*ans_i, *adj_y, adj_time, adj_params)
                        slog = getattr(native, 'ndcn_adjoint_step_log', None) if native is not None and method in (None, 'dopri5') else None
                        aug = odeint(augmented, aug0, torch.stack([t[i], t[i - 1]]), rtol=rtol, atol=atol, method=method,
                                     options=options, **({'step_log': slog} if slog is not None else {}))
                        adj_y = tuple(a[1] for a in aug[n:2 * n])
                        adj_time = aug[2 * n][1]
                        adj_params = aug[2 * n + 1][1]
                    if native is not None:
                        adj_y = (hip.combine(adj_y[0].contiguous(), [grad_output[0][i - 1].contiguous()], [1.0]),)
                    else:
                        adj_y = tuple(a + g[i - 1] for a, g in zip(adj_y, grad_output))
                    del aug0, aug
                time_vjps.append(adj_time.reshape(1))
                time_vjps = torch.cat([v.reshape(1) for v in time_vjps[::-1]])
        finally:
            if restore is not None:
                restore[0].ndcn_set_aten_norm_max(restore[1])
        return (None, None, None, time_vjps, adj_params, None, None, None, None) + tuple(adj_y)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None):
    """Same signature, defaults and errors as the reference (adjoint.py:105-133)."""
    if not isinstance(func, nn.Module):
        raise ValueError('func is required to be an instance of nn.Module.')
    tensor_input = torch.is_tensor(y0)
    user_func = None
    if tensor_input:
        user_func = func
        func = _TupleFunc(func)
        y0 = (y0,)
    flat_params = _flat(func.parameters())
    ys = _AdjointMethod.apply(func, user_func, len(y0), t, flat_params, rtol, atol, method, options, *y0)
    return ys[0] if tensor_input else ys
