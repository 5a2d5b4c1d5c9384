"""Host logic of the integrator, written against a small panel-ops interface (ndcn_amd.ops.HipOps).

Mirrors the control flow and scalar arithmetic of the reference's vendored torchdiffeq:
  input checks      torchdiffeq/_impl/misc.py:173-195, odeint.py:61-76
  fixed grid        torchdiffeq/_impl/solvers.py:79-99 ; fixed_grid.py:7-8,17-19,28-29 ; rk_common.py:72-78
  dopri5            torchdiffeq/_impl/dopri5.py:58-122 ; rk_common.py:22-61 ; misc.py:84-170 ; interp.py:5-65
  adams             torchdiffeq/_impl/adams.py:11-170 (variable-coefficient Adams-Bashforth-Moulton)
All per-element work is delegated to `ops` (one fused HIP kernel per reference op chain); what stays
here is scalar: times and the step-size controller in float64, every quantity the reference forms as a
0-d tensor of the state dtype (dt*beta, stage times, initial-step heuristic, interpolation abscissa)
in numpy float32.  State may be a tensor or a tuple of tensors (misc.py:175-182).
"""
import math
import warnings

import numpy as np
import threading

import torch

f32 = np.float32

# Dormand-Prince 5(4) - dopri5.py:11-36, same rational expressions
DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.]
DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_ERR = [
    35 / 384 - 1951 / 21600,
    0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1. / 60.,
]
DP_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2
]
# dopri5.py:60,72-74: the defaults go through torch.tensor(float) (float32) before widening to float64
SAFETY = float(f32(0.9))
IFACTOR = 10.0
DFACTOR = float(f32(0.2))

FIXED_METHODS = ('euler', 'midpoint', 'rk4')
METHODS = FIXED_METHODS + ('dopri5', 'adams')
# methods the reference lists (odeint.py:8-17) but this build does not provide
UNSUPPORTED = ('explicit_adams', 'fixed_adams', 'tsit5')


def _nan_max(a, b):
    return float('nan') if (math.isnan(a) or math.isnan(b)) else max(a, b)


def _nan_min(a, b):
    return float('nan') if (math.isnan(a) or math.isnan(b)) else min(a, b)


def dt_terms(dt32, coeffs, ks):
    """(scale * x) of misc.py:25 in float32; exact-zero tableau entries contribute an exact zero and are
    dropped (the reference multiplies them out, which only matters for non-finite k)."""
    cs, kk = [], []
    for c, k in zip(coeffs, ks):
        c32 = f32(c)
        if c32 == 0:
            continue
        cs.append(f32(dt32 * c32))
        kk.append(k)
    return kk, cs


class TimeArg:
    """Builds the `t` argument of func(t, y): a 0-d tensor in the state's dtype on the state's device,
    as the reference passes (rk_common.py:45-50; solvers.py:90).  Autonomous right-hand sides (ours)
    never read it, so one cached tensor is reused instead of a host-to-device copy per evaluation."""

    def __init__(self, like, autonomous):
        self.like = like
        self.cached = torch.zeros((), dtype=like.dtype, device=like.device) if autonomous else None

    def __call__(self, value):
        if self.cached is not None:
            return self.cached
        return torch.tensor(float(value), dtype=self.like.dtype, device=self.like.device)


# ---- the time grid on the host, fetched ONCE per odeint call ------------------------------------------------------------------------
# The solvers decide on the host: is t increasing, what are the step sizes.  Asked of a device tensor, every such question is a few
# tiny kernels and a synchronisation - four of them per odeint call before round 5, ~0.1 ms of a 1.9 ms README training step
# (tools/micro/trace_gaps.py: the largest idle gaps of the loop sat behind their copies).  odeint() brackets its work with
# grid_scope(t); inside, host_grid(t) answers from the one copy.  (Per call, never across calls: a tensor may be written between them.)

class _GridScope(threading.local):
    tensor = None
    host = None


_GRID = _GridScope()


class grid_scope:
    def __init__(self, t):
        self.t = t

    def __enter__(self):
        self.prev = (_GRID.tensor, _GRID.host)
        if torch.is_tensor(self.t) and self.t.device.type != 'cpu':
            _GRID.tensor, _GRID.host = self.t, self.t.detach().to('cpu')
        return self

    def __exit__(self, *exc):
        _GRID.tensor, _GRID.host = self.prev
        return False


def host_grid(t):
    """t as a CPU tensor of its own dtype (the call's one copy when t is the grid odeint was given)"""
    if t.device.type == 'cpu':
        return t.detach()
    if _GRID.tensor is t:
        return _GRID.host
    return t.detach().to('cpu')


def check_inputs(func, y0, t):
    """misc.py:173-195.  Returns (tensor_input, func_on_tuples, y0_tuple, t, sign)."""
    tensor_input = False
    if torch.is_tensor(y0):
        tensor_input = True
        y0 = (y0,)
        base = func
        func = lambda tt, y: (base(tt, y[0]),)
    assert isinstance(y0, tuple), 'y0 must be either a torch.Tensor or a tuple'
    for y0_ in y0:
        assert torch.is_tensor(y0_), 'each element must be a torch.Tensor but received {}'.format(type(y0_))
    th = host_grid(t)
    if bool((th[1:] < th[:-1]).all()):
        t = -t
        rev = func
        func = lambda tt, y: tuple(-f_ for f_ in rev(-tt, y))
    for y0_ in y0:
        if not torch.is_floating_point(y0_):
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y0_.type()))
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.type()))
    return tensor_input, func, y0, t


def assert_increasing(t):
    th = host_grid(t)
    assert bool((th[1:] > th[:-1]).all()), 't must be strictly increasing or decrasing'


# ---------------------------------------------------------------------------------------------------
# fixed grid
# ---------------------------------------------------------------------------------------------------

def integrate_fixed(ops, func, y0, t, method, autonomous=False):
    """solvers.py:79-99 with the default grid (grid == t).  Returns a list (per tick) of tuples."""
    assert_increasing(t)
    dtype = y0[0].dtype
    tg = host_grid(t).to(dtype).numpy()          # solvers.py:81: times in the state dtype
    targ = TimeArg(y0[0], autonomous)
    sol = [y0]
    y = y0
    for i in range(len(tg) - 1):
        t0, t1 = tg[i], tg[i + 1]
        dt = t1 - t0                                      # float32 subtraction, as `t1 - t0` on 0-d tensors
        if method == 'euler':
            k1 = func(targ(t0), y)
            y = tuple(ops.fixed_stage(0, y_, k_, dt=dt) for y_, k_ in zip(y, k1))
        elif method == 'midpoint':
            k1 = func(targ(t0), y)
            ym = tuple(ops.fixed_stage(1, y_, k_, dt=dt) for y_, k_ in zip(y, k1))
            k2 = func(targ(t0 + dt / f32(2)), ym)
            y = tuple(ops.fixed_stage(0, y_, k_, dt=dt) for y_, k_ in zip(y, k2))
        else:
            k1 = func(targ(t0), y)
            y2 = tuple(ops.fixed_stage(2, y_, a, dt=dt) for y_, a in zip(y, k1))
            k2 = func(targ(t0 + dt / f32(3)), y2)
            y3 = tuple(ops.fixed_stage(3, y_, a, b, dt=dt) for y_, a, b in zip(y, k1, k2))
            k3 = func(targ(t0 + dt * f32(2) / f32(3)), y3)
            y4 = tuple(ops.fixed_stage(4, y_, a, b, c, dt=dt) for y_, a, b, c in zip(y, k1, k2, k3))
            k4 = func(targ(t0 + dt), y4)
            y = tuple(ops.fixed_stage(5, y_, a, b, c, d, dt=dt) for y_, a, b, c, d in zip(y, k1, k2, k3, k4))
        sol.append(y)
    return sol


# ---------------------------------------------------------------------------------------------------
# dopri5
# ---------------------------------------------------------------------------------------------------

def _numel(ops, t):
    """Element count behind a reduction: the tensor's own, or the GLOBAL count when `ops` reduces across
    ranks (ndcn_amd.sharding.DistOps)."""
    fn = getattr(ops, 'numel', None)
    return fn(t) if fn is not None else t.numel()


def _rms(ops, a, b, y, rtol, atol):
    """misc.py:71-76 on (a - b) / (atol + |y| rtol): float32 norm divided by numel ** 0.5."""
    s, bad = ops.scaled_sumsq(a, b, y, rtol, atol)
    nrm = f32(math.sqrt(s)) if s == s and s >= 0 else f32('nan')
    return f32(nrm / f32(math.sqrt(_numel(ops, a)))), bad


def select_initial_step(ops, func, targ, t0, y0, order, rtol, atol, f0):
    """misc.py:84-143 (Hairer II.4).  Returns (h as python float (a float32 value), non-finite count of y0)."""
    t0 = f32(t0)
    stats0 = [_rms(ops, y, None, y, rtol, atol) for y in y0]
    d0 = [s[0] for s in stats0]
    bad0 = sum(s[1] for s in stats0)
    d1 = [_rms(ops, f, None, y, rtol, atol)[0] for f, y in zip(f0, y0)]
    if float(max(d0)) < 1e-5 or float(max(d1)) < 1e-5:
        h0 = f32(1e-6)
    else:
        h0 = f32(f32(0.01) * max(f32(a / b) for a, b in zip(d0, d1)))
    y1 = tuple(ops.combine(y, [f], [h0]) for y, f in zip(y0, f0))
    f1 = func(targ(f32(t0 + h0)), y1)
    d2 = [f32(_rms(ops, b, a, y, rtol, atol)[0] / h0) for b, a, y in zip(f1, f0, y0)]
    if float(max(d1)) <= 1e-15 and float(max(d2)) <= 1e-15:
        h1 = max(f32(1e-6), f32(h0 * f32(1e-3)))
    else:
        m = max(d1 + d2)                       # list concatenation, as in the reference: max over both
        # `(0.01 / m) ** (1 / (order + 1))` as torch evaluates it on a float32 0-d tensor: python_scalar / tensor is
        # tensor.reciprocal() * scalar (two float32 roundings), and tensor ** python_float runs std::pow in double with the
        # exponent at full double precision, rounded to float32 once
        a = f32(f32(f32(1.0) / m) * f32(0.01))
        h1 = f32(math.pow(float(a), 1. / float(order + 1))) if a == a else f32('nan')
    h100 = f32(f32(100) * h0)
    if np.isnan(h100) or np.isnan(h1):
        return float('nan'), bad0
    return float(min(h100, h1)), bad0


def optimal_step_size(dt, ratio32, safety=SAFETY, ifactor=IFACTOR, dfactor=DFACTOR, order=5):
    """misc.py:160-170 in float64 with the reference's float32-born constants (the exponent 1 / order is born a
    float32 tensor: `torch.tensor(1 / order).to(last_step)`)."""
    if ratio32 == 0:
        return dt * ifactor
    dfac = 1.0 if ratio32 < 1 else dfactor
    er = float(np.sqrt(f32(ratio32)))
    expo = float(f32(1.0 / order))
    factor = _nan_max(1.0 / ifactor, _nan_min(math.pow(er, expo) / safety if er == er else float('nan'), 1.0 / dfac))
    return dt / factor


DOPRI5_OPTIONS = ('first_step', 'safety', 'ifactor', 'dfactor', 'max_num_steps')


def controller_constant(v):
    """dopri5.py:72-74: `_convert_to_tensor(v, dtype=float64)` - a python number passes through torch.tensor (float32)
    first, a tensor keeps its own precision."""
    if torch.is_tensor(v):
        return float(v.detach().to(torch.float64))
    return float(f32(v))


def dopri5_options(options, n_state):
    """Validated solver options shared by every dopri5 path: warns about unknown names exactly as the reference
    (`_handle_unused_kwargs`, misc.py:79-81) and returns the controller constants as float64 python floats."""
    unused = {k: v for k, v in options.items() if k not in DOPRI5_OPTIONS}
    if unused:
        warnings.warn('Dopri5Solver: Unexpected arguments {}'.format(unused))
    return {'first_step': options.get('first_step'),
            'safety': controller_constant(options.get('safety', 0.9)),
            'ifactor': controller_constant(options.get('ifactor', 10.0)),
            'dfactor': controller_constant(options.get('dfactor', 0.2)),
            'max_num_steps': int(options.get('max_num_steps', 2 ** 31 - 1))}


def per_state_tolerance(tol, n_state):
    """dopri5.py:69-70: a scalar tolerance applies to every state tensor, an iterable gives one per tensor."""
    if isinstance(tol, (list, tuple)) or (torch.is_tensor(tol) and tol.dim() > 0):
        tol = [float(v) for v in tol]
        assert len(tol) == n_state, 'one tolerance per state tensor expected'
        return tol
    return [float(tol)] * n_state


class Dopri5:
    """dopri5.py:58-122 as a resumable object: begin() = before_integrate, advance() = advance."""

    def __init__(self, ops, func, y0, rtol, atol, autonomous=False, max_num_steps=2 ** 31 - 1, first_step=None,
                 fused=None, safety=SAFETY, ifactor=IFACTOR, dfactor=DFACTOR):
        """`fused`: optional object with rhs_rk(x, mode, y0, kprev, cs, rtol, atol) -> (k, y_next | (sum, bad)):
        the right-hand side that also performs the stage algebra consuming its result (ndcn_rhs_rk_f32).
        Single-tensor states only; the step then costs 1 combine + 6 fused evaluations."""
        self.ops, self.func = ops, func
        self.fused = fused if len(y0) == 1 else None
        self.y = y0
        self.rtol, self.atol = per_state_tolerance(rtol, len(y0)), per_state_tolerance(atol, len(y0))
        self.safety, self.ifactor, self.dfactor = safety, ifactor, dfactor
        self.max_num_steps = max_num_steps
        self.first_step = first_step
        self.targ = TimeArg(y0[0], autonomous)
        self.n_elem = [_numel(ops, y) for y in y0]
        self.log = []                 # (t0, dt, accepted, mean_sq_error_ratio, dt_next) per attempt
        self.nfe = 0

    def _f(self, tval, y):
        self.nfe += 1
        return self.func(self.targ(tval), y)

    def begin(self, t0):
        self.t0 = self.t1 = float(t0)
        self.f = self._f(f32(t0), self.y)
        if self.first_step is None:
            # dopri5.py:80: the initial step is chosen with the FIRST tensor's tolerances
            h, bad = select_initial_step(self.ops, lambda tt, yy: self._count(tt, yy), self.targ, t0, self.y, 4,
                                         self.rtol[0], self.atol[0], self.f)
        else:
            h, bad = 0.01, 0          # dopri5.py:82: a supplied first_step is ignored, 0.01 is used
        self.dt = h
        self.pending_bad = bad
        self.fit = None               # (a, b, c, d, e) tuples-of-tensors of the last fitted step
        self.ticks_in_step = 0        # dense-output evaluations already served from the current accepted step
        self.stage = None             # (y0, y1, k) of the last accepted, not yet fitted step

    def _count(self, tt, yy):
        self.nfe += 1
        return self.func(tt, yy)

    def step(self):
        """dopri5.py:94-122, one attempt."""
        ops = self.ops
        t0, dt = self.t1, self.dt
        assert t0 + dt > t0, 'underflow in dt {}'.format(dt)
        assert self.pending_bad == 0, 'non-finite values in state `y`: {} elements'.format(int(self.pending_bad))
        dt32 = f32(dt)
        t032 = f32(t0)
        y0 = self.y
        if self.fused is not None:
            return self._step_fused(t0, dt, dt32)
        k = [[f] for f in self.f]
        yi = y0
        for a_i, b_i in zip(DP_ALPHA, DP_BETA):
            ti = f32(t032 + f32(f32(a_i) * dt32))
            new = []
            for y_, k_ in zip(y0, k):
                kk, cs = dt_terms(dt32, b_i, k_)
                new.append(ops.combine(y_, kk, cs))
            yi = tuple(new)
            for k_, f_ in zip(k, self._f(ti, yi)):
                k_.append(f_)
        y1 = yi
        ratios, bad_total = [], 0
        for y0_, y1_, k_, n, rtol_, atol_ in zip(y0, y1, k, self.n_elem, self.rtol, self.atol):
            kk, cs = dt_terms(dt32, DP_C_ERR, k_)
            s, bad = ops.error(y0_, y1_, kk, cs, rtol_, atol_)
            ratios.append(f32(s / n) if n else f32('nan'))        # misc.py:156: a float32 mean
            bad_total += bad
        return self._finish_step(t0, dt, dt32, y0, y1, k, ratios, bad_total)

    def _finish_step(self, t0, dt, dt32, y0, y1, k, ratios, bad_total):
        accept = all(bool(r <= 1) for r in ratios)                 # dopri5.py:109
        worst = f32('nan') if any(np.isnan(r) for r in ratios) else max(ratios)
        dt_next = optimal_step_size(dt, worst, self.safety, self.ifactor, self.dfactor)
        self.log.append((t0, dt, 1.0 if accept else 0.0, float(worst), dt_next))
        if accept:
            self.stage = (y0, y1, k, dt32)
            self.fit = None
            self.ticks_in_step = 0
            self.y = y1
            self.f = tuple(k_[-1] for k_ in k)
            self.t0, self.t1 = t0, t0 + dt
            self.pending_bad = bad_total
        else:
            self.t0 = self.t1 = t0
        self.dt = dt_next
        return accept

    def _step_fused(self, t0, dt, dt32):
        """One attempt with the stage algebra riding in the RHS epilogues (same terms, same order as step())."""
        y0 = self.y[0]
        k = [self.f[0]]
        kk, cs = dt_terms(dt32, DP_BETA[0], k)
        x = self.ops.combine(y0, kk, cs)
        aux = bool(getattr(self.fused, 'supports_aux', False))
        E = P = None
        for i in range(1, 6):
            # the evaluation producing k[i] also forms the input of stage i+1: y0 + dt * sum beta[i][m] k[m]
            prev, cp = dt_terms(dt32, DP_BETA[i][:i], k)
            c_new = f32(dt32 * f32(DP_BETA[i][i]))
            self.nfe += 1
            if i == 4 and P is not None:
                prev, cp = [P], [f32(1.0)]                         # y0 + (1 * P + dt beta_65 k5): the same roundings
            if i == 3 and aux and len(cp) == 3:
                # holding k1, k2, k3: the first four terms of the stage-6 sum, P = dt sum_{j<=4} beta_6j k_j, handed to the
                # next evaluation, which then reads {y0, P} instead of {y0, k1..k4} (solver.hip: enqueue_attempt)
                _, c6 = dt_terms(dt32, DP_BETA[4][:4], k + [None])
                if len(c6) == 4:
                    k_new, x_next, P = self.fused.rhs_rk(x, 'combine', y0, prev, cp + [c_new], 0.0, 0.0, aux_cs=c6)
                    k.append(k_new)
                    x = x_next
                    continue
            if i == 5 and aux:
                # ... and, holding k1, k3, k4, k5 already, the partial error sum E = dt sum_{j<=6} c_err[j] k_j (the same
                # stages in the same order: beta[5][j] and c_err[j] vanish for the same j), so that the error evaluation
                # reads {y0, E, y1} instead of seven panels (include/ndcn_hip.h: y_aux)
                _, ce = dt_terms(dt32, DP_C_ERR[:5], k)
                assert len(ce) == len(cp)
                k_new, x_next, E = self.fused.rhs_rk(x, 'combine', y0, prev, cp + [c_new], 0.0, 0.0,
                                                     aux_cs=ce + [f32(dt32 * f32(DP_C_ERR[5]))])
            else:
                k_new, x_next = self.fused.rhs_rk(x, 'combine', y0, prev, cp + [c_new], 0.0, 0.0)
            k.append(k_new)
            x = x_next
        y1 = x
        c_new = f32(dt32 * f32(DP_C_ERR[6]))
        if E is not None:
            prev, cp = [E], [f32(1.0)]
        else:
            prev, cp = dt_terms(dt32, DP_C_ERR[:6], k)
        self.nfe += 1
        k_new, (s, bad) = self.fused.rhs_rk(y1, 'error', y0, prev, cp + [c_new], self.rtol[0], self.atol[0])
        k.append(k_new)
        n = self.n_elem[0]
        ratios = [f32(s / n) if n else f32('nan')]
        return self._finish_step(t0, dt, dt32, (y0,), (y1,), [k], ratios, bad)

    def advance(self, next_t, step_budget=None):
        """dopri5.py:85-92: step until t1 >= next_t, then evaluate the dense output at next_t.
        With `step_budget`, stops after that many attempts and returns None if next_t is not reached yet."""
        next_t = float(next_t)
        n_steps = 0
        while next_t > self.t1:
            assert n_steps < self.max_num_steps, 'max_num_steps exceeded ({}>={})'.format(n_steps, self.max_num_steps)
            if step_budget is not None and n_steps >= step_budget:
                return None
            self.step()
            n_steps += 1
        # interp.py:51-65: abscissa and powers in the state dtype
        a0, a1, at = f32(self.t0), f32(self.t1), f32(next_t)
        assert (a0 <= at) and (at <= a1), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(a0, at, a1)
        x = f32(f32(at - a0) / f32(a1 - a0))
        x2 = f32(x * x)
        x3 = f32(x2 * x)
        x4 = f32(x3 * x)
        xp = (x4, x3, x2, x, f32(1))
        if self.fit is None:
            y0, y1, k, dt32 = self.stage
            cmid = [f32(dt32 * f32(c)) for c in DP_C_MID]
            self.ticks_in_step += 1
            if self.ticks_in_step == 1 and hasattr(self.ops, 'interp_direct'):
                # first tick inside this step: fit + evaluate in one pass, coefficients not stored (most steps are
                # sampled at most once); a second tick pays for the stored fit
                return tuple(self.ops.interp_direct(a_, b_, k_, cmid, dt32, xp) for a_, b_, k_ in zip(y0, y1, k))
            fits = [self.ops.interp_fit(a_, b_, k_, cmid, dt32) for a_, b_, k_ in zip(y0, y1, k)]
            self.fit = (fits, y0)
        fits, e = self.fit
        return tuple(self.ops.interp_eval(fit, e_, xp) for fit, e_ in zip(fits, e))


def integrate_dopri5(ops, func, y0, t, rtol, atol, autonomous=False, step_log=None, fused=None, **options):
    """solvers.py:25-33."""
    assert_increasing(t)
    opt = dopri5_options(options, len(y0))
    tt = host_grid(t).to(torch.float64).numpy()
    solver = Dopri5(ops, func, y0, rtol, atol, autonomous=autonomous, max_num_steps=opt['max_num_steps'],
                    first_step=opt['first_step'], fused=fused, safety=opt['safety'], ifactor=opt['ifactor'],
                    dfactor=opt['dfactor'])
    solver.begin(tt[0])
    sol = [y0]
    for i in range(1, len(tt)):
        sol.append(solver.advance(tt[i]))
    if step_log is not None:
        step_log.extend(solver.log)
        step_log.append(('nfe', solver.nfe))
    return sol


# ---------------------------------------------------------------------------------------------------
# adams: variable-coefficient Adams-Bashforth-Moulton, orders 1..12          torchdiffeq/_impl/adams.py
# ---------------------------------------------------------------------------------------------------

# adams.py:11-15
GAMMA_STAR = [
    1, -1 / 2, -1 / 12, -1 / 24, -19 / 720, -3 / 160, -863 / 60480, -275 / 24192, -33953 / 3628800, -0.00789255,
    -0.00678585, -0.00592406, -0.00523669, -0.0046775, -0.00421495, -0.0038269
]
ADAMS_OPTIONS = ('implicit', 'max_order', 'safety', 'ifactor', 'dfactor')


def adams_g_and_beta(prev_t, next_t, k):
    """The scalar half of adams.py:26-50 in float64 (numpy scalars round like torch's float64 0-d tensors): the g
    coefficients g[0..k] and the factors beta[1..k-1] that turn implicit phi_j into explicit ones."""
    curr_t = prev_t[0]
    dt = next_t - prev_t[0]
    g = np.empty(k + 1, dtype=np.float64)
    g[0] = 1.0
    c = 1.0 / np.arange(1, k + 2, dtype=np.float64)
    beta = np.float64(1.0)
    betas = [None]
    for j in range(1, k):
        beta = (next_t - prev_t[j - 1]) / (curr_t - prev_t[j]) * beta
        betas.append(beta)
        c = c[:-1] - c[1:] if j == 1 else c[:-1] - c[1:] * dt / (next_t - prev_t[j - 1])
        g[j] = c[0]
    c = c[:-1] - c[1:] * dt / (next_t - prev_t[k - 1])
    g[k] = c[0]
    return g, betas


class Adams:
    """adams.py:62-170 as a resumable object (begin = before_integrate, advance = advance).  All per-element work goes
    through three panel ops of `ops`: combine (y + sum c_j k_j: predictor, corrector, and - with c = -1 - the phi
    differences), scale (beta * phi) and error (the mean squared error ratio of ONE scaled phi); the g / beta
    coefficients, the order selection and the step-size controller are scalar host code in the reference's dtypes.
    Quirks kept: the state that is carried on (and returned at the ticks) is the PREDICTOR value p_next, not the
    corrected y_next (:170); every tick is a step end (next_t is clipped to it, :105-106) - no dense output."""

    def __init__(self, ops, func, y0, rtol, atol, autonomous=False, max_order=12, safety=SAFETY, ifactor=IFACTOR,
                 dfactor=DFACTOR):
        self.ops, self.func = ops, func
        self.y = y0
        self.rtol, self.atol = per_state_tolerance(rtol, len(y0)), per_state_tolerance(atol, len(y0))
        self.max_order = int(max(1, min(max_order, 12)))
        self.safety, self.ifactor, self.dfactor = safety, ifactor, dfactor
        self.targ = TimeArg(y0[0], autonomous)
        self.n_elem = [_numel(ops, y) for y in y0]
        self.log = []                 # (t_n, attempted next_t, order, accepted, max error ratio, next next_t)
        self.nfe = 0

    def _f(self, tval, y):
        self.nfe += 1
        return self.func(self.targ(tval), y)

    def begin(self, t0):
        t0 = np.float64(t0)
        f0 = self._f(f32(t0), self.y)
        self.prev_t, self.prev_f, self.phi = [t0], [f0], [f0]       # index 0 = newest (the reference's deque.appendleft)
        h, _ = select_initial_step(self.ops, lambda tt, yy: self._count(tt, yy), self.targ, t0, self.y, 2,
                                   self.rtol[0], self.atol[0], f0)
        self.next_t = t0 + np.float64(h)
        self.order = 1

    def _count(self, tt, yy):
        self.nfe += 1
        return self.func(tt, yy)

    def _ratio(self, coef, phis, y0, y1):
        """misc.py:146-157 on coef * phi per state tensor: float32 means."""
        out = []
        for p, a, b, n, rt, at in zip(phis, y0, y1, self.n_elem, self.rtol, self.atol):
            s, _ = self.ops.error(a, b, [p], [coef], rt, at)
            out.append(f32(s / n) if n else f32('nan'))
        return out

    def step(self, final_t):
        ops = self.ops
        y0, prev_t, order = self.y, self.prev_t, self.order
        next_t = final_t if self.next_t > final_t else self.next_t
        assert next_t == next_t, 'adams: step size became NaN'       # (the reference would loop forever here)
        dt = next_t - prev_t[0]
        dt32 = f32(dt)
        g64, betas = adams_g_and_beta(prev_t, next_t, order)
        g = g64.astype(f32)
        # explicit phi (:37-42): phi_0 as is, phi_j scaled by beta_j cast to the state dtype
        ephi = [self.phi[0]] + [tuple(ops.scale(p, f32(betas[j])) for p in self.phi[j]) for j in range(1, order)]
        m = max(1, order - 1)
        cs = [f32(dt32 * g[j]) for j in range(m)]
        p_next = tuple(ops.combine(y_, [ephi[j][q] for j in range(m)], cs) for q, y_ in enumerate(y0))
        nf = self._f(f32(next_t), p_next)
        # implicit phi of the predictor (:53-59): phi_j = phi_{j-1} - explicit phi_{j-1}
        k = min(len(ephi) + 1, order + 1)
        iphi_p = [nf]
        for j in range(1, k):
            iphi_p.append(tuple(ops.combine(a, [b], [f32(-1.0)]) for a, b in zip(iphi_p[j - 1], ephi[j - 1])))
        y_next = tuple(ops.combine(p_, [ip], [f32(dt32 * g[order - 1])]) for p_, ip in zip(p_next, iphi_p[order - 1]))
        error_k = self._ratio(f32(dt32 * f32(g[order] - g[order - 1])), iphi_p[order], y0, y_next)
        accept = all(bool(r <= 1) for r in error_k)
        worst = f32('nan') if any(np.isnan(r) for r in error_k) else max(error_k)
        if not accept:
            dt_next = optimal_step_size(dt, worst, self.safety, self.ifactor, self.dfactor, order=order)
            self.log.append((float(prev_t[0]), float(next_t), order, 0.0, float(worst), float(prev_t[0] + dt_next)))
            self.next_t = prev_t[0] + np.float64(dt_next)
            return False
        nf = self._f(f32(next_t), y_next)
        k = min(len(ephi) + 1, order + 2)
        iphi = [nf]
        for j in range(1, k):
            iphi.append(tuple(ops.combine(a, [b], [f32(-1.0)]) for a, b in zip(iphi[j - 1], ephi[j - 1])))
        next_order = order
        if len(prev_t) <= 4 or order < 3:
            next_order = min(order + 1, 3, self.max_order)
        else:
            e1 = self._ratio(f32(dt32 * f32(g[order - 1] - g[order - 2])), iphi_p[order - 1], y0, y_next)
            e2 = self._ratio(f32(dt32 * f32(g[order - 2] - g[order - 3])), iphi_p[order - 2], y0, y_next)
            if min(e1 + e2) < max(error_k):
                next_order = order - 1
            elif order < self.max_order:
                e3 = self._ratio(f32(dt32 * f32(GAMMA_STAR[order])), iphi_p[order], y0, y_next)
                if max(e3) < max(error_k):
                    next_order = order + 1
        dt_next = dt if next_order > order else optimal_step_size(dt, worst, self.safety, self.ifactor, self.dfactor,
                                                                  order=order + 1)
        self.log.append((float(prev_t[0]), float(next_t), order, 1.0, float(worst), float(next_t + dt_next)))
        self.prev_f = ([nf] + self.prev_f)[:self.max_order + 1]
        self.prev_t = ([next_t] + prev_t)[:self.max_order + 1]
        self.y, self.next_t, self.phi, self.order = p_next, next_t + np.float64(dt_next), iphi[:self.max_order], next_order
        return True

    def advance(self, final_t):
        final_t = np.float64(final_t)
        while final_t > self.prev_t[0]:
            self.step(final_t)
        assert final_t == self.prev_t[0]
        return self.y


def integrate_adams(ops, func, y0, t, rtol, atol, autonomous=False, step_log=None, **options):
    """solvers.py:25-33 for method 'adams'."""
    assert_increasing(t)
    unused = {k: v for k, v in options.items() if k not in ADAMS_OPTIONS}
    if unused:
        warnings.warn('VariableCoefficientAdamsBashforth: Unexpected arguments {}'.format(unused))
    tt = host_grid(t).to(torch.float64).numpy()
    solver = Adams(ops, func, y0, rtol, atol, autonomous=autonomous, max_order=options.get('max_order', 12),
                   safety=controller_constant(options.get('safety', 0.9)),
                   ifactor=controller_constant(options.get('ifactor', 10.0)),
                   dfactor=controller_constant(options.get('dfactor', 0.2)))
    solver.begin(tt[0])
    sol = [y0]
    for i in range(1, len(tt)):
        sol.append(solver.advance(tt[i]))
    if step_log is not None:
        step_log.extend(solver.log)
        step_log.append(('nfe', solver.nfe))
    return sol
