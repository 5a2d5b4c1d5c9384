"""`odeint_adjoint`'s reverse pass for ODEFunc on the fused launches of the inference path (round 5; SURVEY 8f rank 1: "adjoint
gives O(1) memory at 1M nodes where backprop-through-solver cannot fit", reference torchdiffeq/_impl/adjoint.py:23-102).

The augmented system (adjoint.py:34-59) for f(y) = relu(W (A y) + b), integrated from t_i back to t_{i-1} by dopri5 with the
reference's tuple semantics (one error ratio per state tensor, accept iff all <= 1, step size from the worst; dopri5.py:94-122,
misc.py:146-170):

    y'      =  K                         K  = relu(W S + b),  S = A y
    a_y'    = -A^T (gZ W)                gZ = a_y (.) [K > 0]
    a_t'    =  0                         (ODEFunc ignores t)
    a_th'   = (-gZ^T S, -sum_rows gZ)    (weight, bias; zeros for an unused `wt` under no_control)

What the generic path (adjoint.py of this package: the python tuple stepper over `hip.adjoint_rhs`) spends per evaluation - SpMM,
Linear, masked Linear backward (gS, gW, gb), SpMM with A^T, and per stage a separate combine pass per state tensor - becomes per
evaluation:

    launch 1   ndcn_rhs_rk_f32(A, y_stage, W, b)        -> K          + epilogue: the NEXT stage input of y     (rhs_fused3 / 2)
    mask       gZ = a_stage (.) [K > 0]                               (ndcn_relu_bwd_f32)
    launch 2   ndcn_rhs_rk_f32(A^T, gZ, W^T, no bias, no relu) -> A^T gZ W + epilogue: the NEXT stage input of a_y
    wgrad      gZ^T S, sum_rows gZ with S = A y_stage                 (ndcn_spmm_f32 + ndcn_linear_bwd_f32, fixed-order chunk sums)

using  A^T (gZ W) = (A^T gZ) W:  the transposed half is the SAME fused launch as the forward half, on the operator / weight pair
(A^T, W^T) without bias and activation - gather first, dense product on the matrix cores second, stage algebra in the epilogue.
(The reference's autograd forms gZ W first; the two orders differ by fp32 rounding only.)  The seventh evaluation of a step carries
the error records of y and a_y in its two epilogues (`y1` argument: the record is formed over a_y's rows while the launch gathers gZ).

Time runs backwards: the solver integrates in tau = -t (misc.py:184-187 negates t and func); instead of negating K panels the SIGN
rides in the step size of each state tensor's panel algebra (dt * beta * (-K) == (-dt) * beta * K bit for bit):  y uses -dt, a_y and
a_theta (whose stored panels are +A^T gZ W, +gZ^T S: already the negated derivative) use +dt.

Scope: dopri5 - ODEFunc with H = 256 (the fused kernels' width), graph on, dropout inactive; control on or off.  Fixed grids
(round 6: euler / midpoint / rk4, `FusedAdjointFixed` below) - any width with the graph on: the same two launches per evaluation with the
method's stage algebra in their epilogues (`rk4` / `combine` modes of ndcn_rhs_rk_f32; the narrow widths run rhs_small.hip).  Everything
else - dopri5 on narrow panels, whose accept / reject decisions hang on ATen-order norms of every state tensor - keeps the generic
path.  NDCN_ADJOINT_FUSED=0 switches this off (A/B)."""
import math
import os

import numpy as np
import torch

from . import core
from .core import DP_ALPHA, DP_BETA, DP_C_ERR, DP_C_MID, dt_terms, f32

ENABLED = os.environ.get('NDCN_ADJOINT_FUSED', '1') != '0'


def applicable(f0, y):
    """f0: this package's ODEFunc; y: the saved trajectory (T, N, H)"""
    if not ENABLED or f0.no_graph or y.shape[2] != 256 or y.dtype != torch.float32:
        return False
    from ...csr import as_csr
    A = as_csr(f0.A)
    return A.shape[0] == A.shape[1] == y.shape[1]


class _Weights:
    """W, b and the contiguous W^T the transposed launch reads as ITS weight matrix (one tensor object per solve: the packed-image
    cache of ops.py keys on it)."""

    def __init__(self, f0):
        self.no_control = bool(f0.no_control)
        self.W = self.b = self.Wt = None
        if not self.no_control:
            self.W = f0.wt.weight.detach().contiguous()
            self.b = f0.wt.bias.detach().contiguous() if f0.wt.bias is not None else None
            self.Wt = self.W.t().contiguous()


class FusedAdjointDopri5:
    """One tick interval of the reverse pass: begin(tau0) / advance(tau1) in tau = -t."""

    def __init__(self, hip, f0, weights, y, a, a_t, theta, rtol, atol, options, generic_func):
        from ...csr import as_csr
        self.hip = hip
        self.A = as_csr(f0.A)
        self.At = self.A.transpose()
        self.A.ensure_plans(256)
        self.At.ensure_plans(256)
        self.w = weights
        self.generic = generic_func                        # (tau, aug) -> negated augmented derivative (composed kernels): initial step only
        self.Y, self.Aj, self.a_t, self.P = y, a, a_t, theta
        opt = core.dopri5_options(options or {}, 4)
        self.safety, self.ifactor, self.dfactor = opt['safety'], opt['ifactor'], opt['dfactor']
        self.max_num_steps, self.first_step = opt['max_num_steps'], opt['first_step']
        self.rtol, self.atol = core.per_state_tolerance(rtol, 4), core.per_state_tolerance(atol, 4)
        self.has_theta = theta.numel() > 0 and not self.w.no_control
        self.n_theta = theta.numel()
        self.a_t_host = float(a_t)                          # constant over the interval (a_t' = 0)
        # lattice plans (rhs_fused3): S = A x leaves the forward launch, gZ = a (.) [K > 0] is formed on the rows the transposed launch
        # stages (ndcn_rhs_rk_adj_f32) - no SpMM and no mask pass of their own
        ctl = not self.w.no_control
        self.fuse_s = ctl and self.has_theta and hip.rhs_adj_supported(self.A, 256, 'combine', 1)
        self.fuse_mask = ctl and hip.rhs_adj_supported(self.At, 256, 'combine', 1)
        self.log = []
        self.nfe = 0

    # ---- one evaluation of the augmented right-hand side on the fused launches ------------------------------------------------
    def _transposed(self, gZ, mode, y0, prev, cs, rtol=0.0, atol=0.0, y1=None, x_mask=None):
        w = self.w
        if mode is None:
            return self.hip.rhs(self.At, gZ, w.Wt, None, no_control=w.no_control, relu=False), None
        return self.hip.rhs_rk(self.At, gZ, w.Wt, None, mode, y0, prev, cs, rtol, atol, no_control=w.no_control, relu=False, y1=y1,
                               x_mask=x_mask)

    def _forward(self, x, mode, y0, prev, cs, rtol=0.0, atol=0.0, s_out=None):
        w = self.w
        if mode is None:
            return self.hip.rhs(self.A, x, w.W, w.b, no_control=w.no_control), None
        return self.hip.rhs_rk(self.A, x, w.W, w.b, mode, y0, prev, cs, rtol, atol, no_control=w.no_control, s_out=s_out)

    def _theta(self, x, g, mask=None, S=None):
        """((g (.) [mask > 0])^T S, its row sum) flattened like the parameter list [weight, bias]; S = A x unless handed in"""
        if not self.has_theta:
            return None
        if S is None:
            S = self.hip.spmm(self.A, x)
        _, gW, gb = self.hip.linear_bwd(g, self.w.W, S=S, Y=mask, need_gS=False, need_gW=True, need_gb=self.w.b is not None)
        return torch.cat([gW.view(-1), gb]) if gb is not None else gW.view(-1)

    def _evaluate(self, x, xa, mode, Y0, A0, prevY, cY, prevA, cA, tol_y=(0.0, 0.0), tol_a=(0.0, 0.0)):
        """both halves of one evaluation at the stage inputs (x, xa) + the stage algebra / error records in their epilogues"""
        hip = self.hip
        S = torch.empty_like(x) if self.fuse_s else None
        K, ry = self._forward(x, mode, Y0, prevY, cY, tol_y[0], tol_y[1], s_out=S)
        y1 = xa if mode == 'error' else None
        if self.fuse_mask:
            KA, ra = self._transposed(xa, mode, A0, prevA, cA, tol_a[0], tol_a[1], y1=y1, x_mask=K)
            kp = self._theta(x, xa, K, S)
        else:
            gZ = hip.relu_bwd(xa, K)
            KA, ra = self._transposed(gZ, mode, A0, prevA, cA, tol_a[0], tol_a[1], y1=y1)
            kp = self._theta(x, gZ, None, S)
        return K, ry, KA, ra, kp

    # ---- dopri5.py:76-83 ------------------------------------------------------------------------------------------------------
    def begin(self, tau0):
        self.t0 = self.t1 = float(tau0)
        hip = self.hip
        K, _ = self._forward(self.Y, None, None, None, None)
        gZ = hip.relu_bwd(self.Aj, K)
        KA, _ = self._transposed(gZ, None, None, None, None)
        KP = self._theta(self.Y, gZ)
        self.nfe += 1
        self.k1 = (K, KA, KP)
        if self.first_step is None:
            zero_t = torch.zeros_like(self.a_t)
            theta_k = KP if KP is not None else torch.zeros_like(self.P)
            f_signed = (hip.scale(K, -1.0), KA, zero_t, theta_k)          # the negated derivative, as misc.py:184-187 hands it over
            aug = (self.Y, self.Aj, self.a_t, self.P)

            def counted(tt, yy):
                self.nfe += 1
                return self.generic(tt, yy)
            h, bad = core.select_initial_step(hip, counted, core.TimeArg(self.Y, True), tau0, aug, 4, self.rtol[0], self.atol[0], f_signed)
        else:
            h, bad = 0.01, 0
        self.dt = h
        self.pending_bad = bad
        self.stage = None

    # ---- dopri5.py:94-122, one attempt ----------------------------------------------------------------------------------------
    def step(self):
        hip = self.hip
        t0, dt = self.t1, self.dt
        assert t0 + dt > t0, 'underflow in dt {}'.format(dt)
        assert self.pending_bad == 0, 'non-finite values in state `y`: {} elements'.format(int(self.pending_bad))
        dt32 = f32(dt)
        dty = f32(-dt32)                                   # y's stored panels are +K: the sign of the reversed time rides in dt
        Y0, A0, P0 = self.Y, self.Aj, self.P
        kY, kA, kP = [self.k1[0]], [self.k1[1]], [self.k1[2]]
        kk, cs = dt_terms(dty, DP_BETA[0], kY)
        x = hip.combine(Y0, kk, cs)
        kk, cs = dt_terms(dt32, DP_BETA[0], kA)
        xa = hip.combine(A0, kk, cs)
        for i in range(1, 6):
            prevY, cY = dt_terms(dty, DP_BETA[i][:i], kY)
            prevA, cA = dt_terms(dt32, DP_BETA[i][:i], kA)
            self.nfe += 1
            K, x_next, KA, xa_next, kp = self._evaluate(x, xa, 'combine', Y0, A0, prevY, cY + [f32(dty * f32(DP_BETA[i][i]))],
                                                        prevA, cA + [f32(dt32 * f32(DP_BETA[i][i]))])
            kP.append(kp)
            kY.append(K)
            kA.append(KA)
            x, xa = x_next, xa_next
        y1, a1 = x, xa
        prevY, cY = dt_terms(dty, DP_C_ERR[:6], kY)
        prevA, cA = dt_terms(dt32, DP_C_ERR[:6], kA)
        self.nfe += 1
        K, (sY, badY), KA, (sA, badA), kp = self._evaluate(y1, a1, 'error', Y0, A0, prevY, cY + [f32(dty * f32(DP_C_ERR[6]))],
                                                           prevA, cA + [f32(dt32 * f32(DP_C_ERR[6]))], (self.rtol[0], self.atol[0]),
                                                           (self.rtol[1], self.atol[1]))
        kP.append(kp)
        kY.append(K)
        kA.append(KA)
        n = Y0.numel()
        ratios = [f32(sY / n), f32(sA / n)]
        # a_t: derivative 0 -> error 0 -> ratio 0 (NaN state -> NaN tolerance -> NaN, as the reference's 0 / nan)
        at = self.a_t_host
        ratios.append(f32(0.0) if math.isfinite(at) else f32('nan'))
        bad = badY + badA + (0 if math.isfinite(at) else 1)
        p1 = P0
        if self.has_theta:
            kk, cs = dt_terms(dt32, DP_BETA[5], kP[:6])
            p1 = hip.combine(P0, kk, cs)
            kk, cs = dt_terms(dt32, DP_C_ERR, kP)
            sP, badP = hip.error(P0, p1, kk, cs, self.rtol[3], self.atol[3])
            ratios.append(f32(sP / self.n_theta))
            bad += badP
        elif self.n_theta:
            ratios.append(f32(0.0))
        accept = all(bool(r <= 1) for r in ratios)                 # dopri5.py:109
        worst = f32('nan') if any(np.isnan(r) for r in ratios) else max(ratios)
        dt_next = core.optimal_step_size(dt, worst, self.safety, self.ifactor, self.dfactor)
        self.log.append((t0, dt, 1.0 if accept else 0.0, float(worst), dt_next))
        if accept:
            self.stage = ((Y0, A0, P0), (y1, a1, p1), (kY, kA, kP), dt32)
            self.Y, self.Aj, self.P = y1, a1, p1
            self.k1 = (kY[-1], kA[-1], kP[-1])
            self.t0, self.t1 = t0, t0 + dt
            self.pending_bad = bad
        else:
            self.t0 = self.t1 = t0
        self.dt = dt_next
        return accept

    # ---- dopri5.py:85-92 + interp.py:38-65 -------------------------------------------------------------------------------------
    def advance(self, next_t):
        next_t = float(next_t)
        n_steps = 0
        while next_t > self.t1:
            assert n_steps < self.max_num_steps, 'max_num_steps exceeded ({}>={})'.format(n_steps, self.max_num_steps)
            self.step()
            n_steps += 1
        a0, a1, at = f32(self.t0), f32(self.t1), f32(next_t)
        assert (a0 <= at) and (at <= a1), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(a0, at, a1)
        if self.stage is None:                                      # next_t == tau0: nothing integrated
            return self.Y, self.Aj, self.a_t, self.P
        x = f32(f32(at - a0) / f32(a1 - a0))
        x2 = f32(x * x)
        x3 = f32(x2 * x)
        x4 = f32(x3 * x)
        xp = (x4, x3, x2, x, f32(1))
        (Y0, A0, P0), (y1, a1_, p1), (kY, kA, kP), dt32 = self.stage
        dty = f32(-dt32)
        hip = self.hip
        out_y = hip.interp_direct(Y0, y1, kY, [f32(dty * f32(c)) for c in DP_C_MID], dty, xp)
        cmid = [f32(dt32 * f32(c)) for c in DP_C_MID]
        out_a = hip.interp_direct(A0, a1_, kA, cmid, dt32, xp)
        out_p = hip.interp_direct(P0, p1, kP, cmid, dt32, xp) if self.has_theta else P0
        return out_y, out_a, self.a_t, out_p


def applicable_fixed(f0, y):
    """the fixed-grid reverse pass: any width, graph on, a square operator over the state's rows"""
    if not ENABLED or f0.no_graph or y.dtype != torch.float32:
        return False
    from ...csr import as_csr
    A = as_csr(f0.A)
    return A.shape[0] == A.shape[1] == y.shape[1]


class FusedAdjointFixed(FusedAdjointDopri5):
    """One tick interval of the reverse pass by ONE step of euler / midpoint / rk4 (the reference's fixed grid is the time vector itself:
    solvers.py:79-99 on [t_i, t_{i-1}], negated by misc.py:184-187), in tau = -t.  The generic path runs `core.integrate_fixed` over the
    4-tuple with one `ndcn_adjoint_rhs_f32` (4 kernels) and one stage kernel per state tensor per stage; here an evaluation is the two
    launches of `_evaluate` with the stage algebra of y and a_y in their epilogues (fixed_grid.py:7-29, rk_common.py:72-78: the operator
    order of ndcn_fixed_stage_f32), the parameter adjoint's by `fixed_stage` on its 65 k-element vector."""

    def __init__(self, hip, f0, weights, y, a, a_t, theta, generic_func):
        from ...csr import as_csr
        self.hip = hip
        self.A = as_csr(f0.A)
        self.At = self.A.transpose()
        H = y.shape[1]
        self.A.ensure_plans(H)
        self.At.ensure_plans(H)
        self.w = weights
        self.generic = generic_func
        self.Y, self.Aj, self.a_t, self.P = y, a, a_t, theta
        self.has_theta = theta.numel() > 0 and not self.w.no_control
        self.n_theta = theta.numel()
        self.fuse_s = self.fuse_mask = False               # (the masked-input / S-output kernel variants are dopri5's launches)
        self.nfe = 0

    def step(self, method, dt32):
        hip = self.hip
        dty = f32(-dt32)                                   # y's stored panels are +K: the sign of the reversed time rides in dt
        Y0, A0, P0 = self.Y, self.Aj, self.P
        if method == 'euler':
            self.nfe += 1
            _, y1, _, a1, kp = self._evaluate(Y0, A0, 'combine', Y0, A0, [], [dty], [], [dt32])
            p1 = hip.fixed_stage(0, P0, kp, dt=dt32) if self.has_theta else P0
        elif method == 'midpoint':
            self.nfe += 2
            _, ym, _, am, _ = self._evaluate(Y0, A0, 'combine', Y0, A0, [], [f32(dty / f32(2))], [], [f32(dt32 / f32(2))])
            _, y1, _, a1, kp = self._evaluate(ym, am, 'combine', Y0, A0, [], [dty], [], [dt32])
            p1 = hip.fixed_stage(0, P0, kp, dt=dt32) if self.has_theta else P0
        else:
            x, xa, kY, kA, kP = Y0, A0, [], [], []
            for _ in range(4):
                self.nfe += 1
                K, x, KA, xa, kp = self._evaluate(x, xa, 'rk4', Y0, A0, list(kY), [dty], list(kA), [dt32])
                kY.append(K)
                kA.append(KA)
                kP.append(kp)
            y1, a1 = x, xa
            p1 = hip.fixed_stage(5, P0, kP[0], kP[1], kP[2], kP[3], dt=dt32) if self.has_theta else P0
        self.Y, self.Aj, self.P = y1, a1, p1
        return y1, a1, self.a_t, p1


def integrate_interval_fixed(hip, f0, weights, y_i, adj_y, adj_time, adj_params, t_hi, t_lo, method, generic_func, step_log=None):
    """(y, a_y, a_t, a_theta) at t_hi -> at t_lo < t_hi by one step of `method`; t_hi / t_lo: host scalars of the solve's time vector
    (the step size is their float32 difference in the negated grid, solvers.py:81)."""
    s = FusedAdjointFixed(hip, f0, weights, y_i, adj_y, adj_time, adj_params, generic_func)
    dt32 = f32(f32(-t_lo) - f32(-t_hi))
    out = s.step(method, dt32)
    if step_log is not None:
        step_log.append(('nfe', s.nfe))
    return out


def integrate_interval(hip, f0, weights, y_i, adj_y, adj_time, adj_params, t_hi, t_lo, rtol, atol, options, generic_func, step_log=None):
    """(y, a_y, a_t, a_theta) at t_hi -> at t_lo < t_hi."""
    s = FusedAdjointDopri5(hip, f0, weights, y_i, adj_y, adj_time, adj_params, rtol, atol, options, generic_func)
    s.begin(-float(t_hi))
    out = s.advance(-float(t_lo))
    if step_log is not None:
        step_log.extend(s.log)
        step_log.append(('nfe', s.nfe))
    return out
