"""CPU oracle for the NDCN ODEFunc hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on torch-CPU / numpy, the arithmetic the reference performs on the path
`odeint(ODEFunc, x0, t)`; it exists so the HIP path can be checked against something that is itself
pinned to the reference.  Nothing under `ndcn_amd/` may import it: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and there only as the checker
/ the reported CPU baseline (kind "port").

Parity pin: every public function here is compared in `tests/test_oracle_golden.py` with the fixtures
under `tests/golden/`, which `tools/gen_golden.py` produced by running the reference itself
(/root/reference, torch 2.10.0 CPU).  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), and its arithmetic lives in PyTorch ATen, whose version the reference leaves
unpinned - so "pinned" means: pinned to the reference's Python as executed on torch 2.10.0.

The same ATen primitives the reference calls are used (torch.sparse.mm on a COO operator,
nn.functional.linear, relu) and the op ORDER of the integrator is followed term by term (coefficients
rounded to the state dtype as `dt*beta` before multiplying, left-to-right sums, no FMA), so the
comparison with the fixtures is tight (<= 1e-6; bit-exact in practice).

Reference map (all paths relative to /root/reference):
  rhs                       neural_dynamics.py:20-39
  fixed grid loop           torchdiffeq/_impl/solvers.py:79-99
  euler / midpoint / rk4    torchdiffeq/_impl/fixed_grid.py:7-8,17-19,28-29 ; rk_common.py:72-78 (3/8 rule)
  dopri5 tableau            torchdiffeq/_impl/dopri5.py:11-36
  rk step                   torchdiffeq/_impl/rk_common.py:22-61 ; misc.py:22-25
  initial step              torchdiffeq/_impl/misc.py:84-143
  error ratio / controller  torchdiffeq/_impl/misc.py:146-170 ; dopri5.py:94-122
  dense output              torchdiffeq/_impl/dopri5.py:39-45 ; interp.py:5-65
  input checks              torchdiffeq/_impl/misc.py:173-195 ; odeint.py:61-76
  operators / grid / x0     utils_in_learn_dynamics.py:80-157 ; heat_dynamics.py:116-117,178-182
  truth dynamics            heat_dynamics.py:186-204 ; gene_dynamics.py:186-205 ; mutualistic_dynamics.py:186-232
  zipf alpha operator       propagation.py:91-103
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# operator containers
# ------------------------------------------------------------------------------------------------


def coo_from_csr(indptr, indices, data, shape):
    """torch sparse COO (row-major entry order, as dense.nonzero() gives: utils_in_learn_dynamics.py:193-201)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(len(indptr) - 1, dtype=np.int64), np.diff(indptr))
    idx = torch.from_numpy(np.vstack([rows, np.asarray(indices, dtype=np.int64)]))
    return torch.sparse_coo_tensor(idx, torch.as_tensor(np.asarray(data, dtype=np.float32)),
                                   tuple(int(s) for s in shape))


def dense_from_csr(indptr, indices, data, shape):
    m = sp.csr_matrix((np.asarray(data), np.asarray(indices), np.asarray(indptr)), shape=tuple(int(s) for s in shape))
    return torch.from_numpy(m.toarray().astype(np.float32))


def apply_operator(A, x):
    # neural_dynamics.py:28-31: sparse operators go through torch.sparse.mm, dense ones through torch.mm
    if getattr(A, 'is_sparse', False):
        return torch.sparse.mm(A, x)
    return torch.mm(A, x)


# ------------------------------------------------------------------------------------------------
# the right-hand side
# ------------------------------------------------------------------------------------------------


def odefunc_rhs(A, x, W, b, no_graph=False, no_control=False):
    """relu(dropout_0(W (A x) + b)); neural_dynamics.py:27-36 with dropout p=0 (the README setting)."""
    if not no_graph:
        x = apply_operator(A, x)
    if not no_control:
        x = F.linear(x, W, b)
    return F.relu(x)


class OracleODEFunc:
    """Callable (t, x) -> dx/dt with the reference ODEFunc's semantics; counts evaluations."""

    def __init__(self, A, W, b, no_graph=False, no_control=False):
        self.A, self.W, self.b = A, W, b
        self.no_graph, self.no_control = no_graph, no_control
        self.nfe = 0

    def __call__(self, t, x):
        self.nfe += 1
        return odefunc_rhs(self.A, x, self.W, self.b, self.no_graph, self.no_control)


def gcn_layer(A, x, W, b):
    """GraphConvolution: A (x W^T + b)  (models.py:14-18; neural_dynamics.py:170-176 before the view)."""
    return apply_operator(A, F.linear(x, W, b))


def row_normalization(X):
    """ode_gcn.py:9-16 / RowNorm ode_gcn.py:19-26: each row divided by max(its L1 norm, 1e-12), infinities zeroed."""
    X = F.normalize(X.float(), 1, 1)
    X[torch.isinf(X)] = 0
    return X


def resblock(A, x, W=None, b=None, time_step=1.0, normalize=False):
    """ResBlock.forward ode_gcn.py:48-60, dropout 0: x + relu([rownorm]([W]((A [rownorm]x)))) * time_step."""
    shortcut = x
    if normalize:
        x = row_normalization(x)
    f = torch.sparse.mm(A, x) if A.is_sparse else torch.mm(A, x)
    if W is not None:
        f = F.linear(f, W, b)
    if normalize:
        f = row_normalization(f)
    return shortcut + F.relu(f) * time_step


def gcn_forward(sd, A, x, n_middle=0):
    """models.GCN.forward (models.py:33-47) in eval mode: gc1 -> relu -> [middle -> relu]* -> gc2, every layer
    A (x W^T + b) (models.py:14-18).  `sd` = the reference state_dict."""
    x = F.relu(gcn_layer(A, x, sd['gc1.fc.weight'], sd['gc1.fc.bias']))
    for i in range(n_middle):
        x = F.relu(gcn_layer(A, x, sd['conv_middle.%d.fc.weight' % i], sd['conv_middle.%d.fc.bias' % i]))
    return gcn_layer(A, x, sd['gc2.fc.weight'], sd['gc2.fc.bias'])


def resgcn_forward(sd, A, x, n_blocks, normalize=False):
    """dgnn.py:129-140 `resGCN` Sequential in eval mode: Linear -> ReLU -> ResBlock x n -> Linear."""
    x = F.relu(F.linear(x, sd['0.weight'], sd['0.bias']))
    for i in range(n_blocks):
        ts = sd.get('%d.time_step' % (2 + i))
        x = resblock(A, x, time_step=1.0 if ts is None else ts, normalize=normalize)
    k = 2 + n_blocks
    return F.linear(x, sd['%d.weight' % k], sd['%d.bias' % k])


# ------------------------------------------------------------------------------------------------
# truth dynamics (ground-truth generators of the three drivers)
# ------------------------------------------------------------------------------------------------


def heat_rhs(L, x, k=1.0):
    """heat_dynamics.py:189-204: the module stores -L and returns k * ((-L) x)."""
    return k * apply_operator(-L if not L.is_sparse else (-1.0) * L, x)


def gene_rhs(A, x, b=1.0, f=1, h=2):
    """gene_dynamics.py:194-205."""
    return -b * (x ** f) + apply_operator(A, x ** h / (x ** h + 1))


def mutual_rhs(A, x, b=0.1, k=5., c=1., d=5., e=0.9, h=0.1):
    """mutualistic_dynamics.py:205-216, the branch that EXECUTES for an N x 1 state: the coupling
    term is sum_j A_ij x_i x_j / (d + e*x_j + h*x_i) (e, h swapped w.r.t. the docstring)."""
    n = x.shape[0]
    out = b + x * (1 - x / k) * (x / c - 1)
    M = torch.mm(x, x.t()) / (d + (e * x).repeat(1, n) + (h * x.t()).repeat(n, 1))
    Ad = A.to_dense() if A.is_sparse else A
    # diag(A @ M)[i] = sum_j A_ij M_ji
    out = out + torch.diag(torch.mm(Ad, M)).view(-1, 1)
    return out


def mutual_rhs_edgewise(indptr, indices, data, x, b=0.1, k=5., c=1., d=5., e=0.9, h=0.1):
    """O(nnz) form of `mutual_rhs` (same formula per stored edge, float64 row sums) for sizes where the
    dense N x N intermediate of the reference cannot exist."""
    xv = np.asarray(x, dtype=np.float64).reshape(-1)
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    cols = np.asarray(indices, dtype=np.int64)
    # M_ji with j=col, i=row:  x_j x_i / (d + e x_j + h x_i)
    term = np.asarray(data, dtype=np.float64) * xv[cols] * xv[rows] / (d + e * xv[cols] + h * xv[rows])
    acc = np.zeros_like(xv)
    np.add.at(acc, rows, term)
    return (b + xv * (1 - xv / k) * (xv / c - 1) + acc).reshape(-1, 1)


# ------------------------------------------------------------------------------------------------
# integrator
# ------------------------------------------------------------------------------------------------

# Dormand-Prince 5(4), dopri5.py:11-31 (values written as the same rational expressions)
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
# dopri5.py:33-36
DP_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2
]
# dopri5.py:60,72-74: the defaults pass through torch.tensor(python_float) - a float32 tensor - before
# being widened to float64, so the controller really uses float32(0.9) and float32(0.2)
SAFETY = float(np.float32(0.9))
IFACTOR = 10.0
DFACTOR = float(np.float32(0.2))


def _wsum(scale, coeffs, tensors):
    """misc.py:22-25: sum_j (scale*c_j) * k_j, scale a 0-d tensor in the state dtype, terms added left
    to right starting from the integer 0; zero coefficients are NOT skipped (k_j is a tensor)."""
    acc = 0
    for c, k in zip(coeffs, tensors):
        acc = acc + (scale * c) * k
    return acc


def _dot(coeffs, tensors):
    """misc.py:28-30."""
    acc = 0
    for c, k in zip(coeffs, tensors):
        acc = acc + c * k
    return acc


def _rms(x):
    """misc.py:71-76 for a single tensor."""
    return x.norm() / (x.numel() ** 0.5)


def _as_state(y0):
    if torch.is_tensor(y0):
        return True, [y0]
    assert isinstance(y0, tuple), 'y0 must be either a torch.Tensor or a tuple'
    for y in y0:
        assert torch.is_tensor(y), 'each element must be a torch.Tensor but received {}'.format(type(y))
    return False, list(y0)


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, step_log=None):
    """Restatement of odeint.py:20-76 for methods dopri5 / euler / midpoint / rk4 / adams.

    `step_log`, when a list, receives one (t0, dt, accepted, mean_sq_error_ratio, dt_next) row per
    attempted dopri5 step - the same quantities tools/gen_golden.py records from the reference.
    """
    single, state = _as_state(y0)
    user_func = func
    if single:
        fn = lambda tt, ys: [user_func(tt, ys[0])]
    else:
        fn = lambda tt, ys: list(user_func(tt, tuple(ys)))
    # misc.py:184-187 - a decreasing time vector is integrated as -t with the sign of f flipped
    if bool((t[1:] < t[:-1]).all()):
        t = -t
        inner = fn
        fn = lambda tt, ys: [-f for f in inner(-tt, ys)]
    for y in state:
        if not torch.is_floating_point(y):
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y.type()))
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.type()))
    if options is None:
        options = {}
    elif method is None:
        raise ValueError('cannot supply `options` without specifying `method`')
    if method is None:
        method = 'dopri5'
    assert bool((t[1:] > t[:-1]).all()), 't must be strictly increasing or decrasing'
    if method == 'dopri5':
        sol = _dopri5(fn, state, t, rtol, atol, step_log, **{k: v for k, v in options.items()
                                                             if k in ('safety', 'ifactor', 'dfactor')})
    elif method in ('euler', 'midpoint', 'rk4'):
        sol = _fixed_grid(fn, state, t, method)
    elif method == 'adams':
        sol = _adams(fn, state, t, rtol, atol, step_log, **{k: v for k, v in options.items()
                                                           if k in ('max_order', 'safety', 'ifactor', 'dfactor')})
    else:
        raise KeyError(method)
    out = [torch.stack([s[i] for s in sol]) for i in range(len(state))]
    return out[0] if single else tuple(out)


def _fixed_grid(fn, y, t, method):
    # solvers.py:79-99 with the default grid (grid == t): every tick is a step end, so the
    # interpolation at :96 returns y1 (:104-105)
    t = t.type_as(y[0])
    sol = [y]
    for t0, t1 in zip(t[:-1], t[1:]):
        dt = t1 - t0
        if method == 'euler':
            dy = [dt * f for f in fn(t0, y)]                                   # fixed_grid.py:8
        elif method == 'midpoint':
            y_mid = [y_ + f_ * dt / 2 for y_, f_ in zip(y, fn(t0, y))]          # fixed_grid.py:18
            dy = [dt * f for f in fn(t0 + dt / 2, y_mid)]                        # fixed_grid.py:19
        else:                                                                    # rk_common.py:72-78
            k1 = fn(t0, y)
            k2 = fn(t0 + dt / 3, [y_ + dt * a / 3 for y_, a in zip(y, k1)])
            k3 = fn(t0 + dt * 2 / 3, [y_ + dt * (a / -3 + b) for y_, a, b in zip(y, k1, k2)])
            k4 = fn(t0 + dt, [y_ + dt * (a - b + c) for y_, a, b, c in zip(y, k1, k2, k3)])
            dy = [(a + 3 * b + 3 * c + d) * (dt / 8) for a, b, c, d in zip(k1, k2, k3, k4)]
        y = [y_ + d_ for y_, d_ in zip(y, dy)]
        sol.append(y)
    return sol


def _initial_step(fn, t0, y0, order, rtol, atol, f0):
    # misc.py:118-143
    t0 = t0.to(y0[0])
    scale = [atol + torch.abs(y) * rtol for y in y0]
    d0 = [_rms(y / s) for y, s in zip(y0, scale)]
    d1 = [_rms(f / s) for f, s in zip(f0, scale)]
    if max(d0).item() < 1e-5 or max(d1).item() < 1e-5:
        h0 = torch.tensor(1e-6).to(t0)
    else:
        h0 = 0.01 * max(a / b for a, b in zip(d0, d1))
    y1 = [y + h0 * f for y, f in zip(y0, f0)]
    f1 = fn(t0 + h0, y1)
    d2 = [_rms((b - a) / s) / h0 for b, a, s in zip(f1, f0, scale)]
    if max(d1).item() <= 1e-15 and max(d2).item() <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6).to(h0), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1 + d2)) ** (1. / float(order + 1))     # list concatenation: max over d1 and d2
    return torch.min(100 * h0, h1)


def _is_finite(x):
    return not bool(((x == float('inf')) + (x == float('-inf')) + torch.isnan(x)).any())


def _dopri5(fn, y_init, t, rtol, atol, step_log, safety=0.9, ifactor=10.0, dfactor=0.2):
    dtype = y_init[0].dtype
    # dopri5.py:72-74: python floats pass through torch.tensor (float32) before widening to float64
    SAFETY, IFACTOR, DFACTOR = (float((v if torch.is_tensor(v) else torch.tensor(v)).type(torch.float64))
                                for v in (safety, ifactor, dfactor))
    t = t.to(torch.float64)                                                   # solvers.py:28
    # dopri5.py:77-83
    f_cur = fn(t[0].type_as(y_init[0]), y_init)
    dt = _initial_step(fn, t[0], y_init, 4, rtol, atol, f_cur).to(t)
    y_cur = y_init
    t_lo = t_hi = t[0]
    coeff = None
    sol = [y_init]
    for i in range(1, len(t)):
        nxt = t[i]
        while nxt > t_hi:                                                     # dopri5.py:88
            t0 = t_hi
            assert t0 + dt > t0, 'underflow in dt {}'.format(dt.item())       # dopri5.py:100
            for y in y_cur:
                assert _is_finite(torch.abs(y)), 'non-finite values in state `y`: {}'.format(y)
            # --- rk_common.py:41-61
            t0s, dts = t0.type(dtype), dt.type(dtype)
            k = [[f] for f in f_cur]
            yi = y_cur
            for a_i, b_i in zip(DP_ALPHA, DP_BETA):
                ti = t0s + a_i * dts
                yi = [y_ + _wsum(dts, b_i, k_) for y_, k_ in zip(y_cur, k)]
                for k_, f_ in zip(k, fn(ti, yi)):
                    k_.append(f_)
            y1 = yi                                                           # c_sol == beta[-1] (:54-58)
            f1 = [k_[-1] for k_ in k]
            err = [_wsum(dts, DP_C_ERR, k_) for k_ in k]
            # --- misc.py:146-157
            ratios = []
            for e_, a_, b_ in zip(err, y_cur, y1):
                tol = atol + rtol * torch.max(torch.abs(a_), torch.abs(b_))
                r = e_ / tol
                ratios.append(torch.mean(r * r))
            accept = bool((torch.tensor(ratios) <= 1).all())                  # dopri5.py:109
            # --- misc.py:160-170
            worst = max(ratios)
            if worst == 0:
                dt_next = dt * IFACTOR
            else:
                dfac = 1.0 if worst < 1 else DFACTOR
                er = torch.sqrt(worst).to(dt)
                expo = torch.tensor(1 / 5).to(dt)
                factor = torch.max(torch.tensor(1 / IFACTOR, dtype=torch.float64),
                                   torch.min(er ** expo / SAFETY, torch.tensor(1 / dfac, dtype=torch.float64)))
                dt_next = dt / factor
            if step_log is not None:
                step_log.append((float(t0), float(dt), 1.0 if accept else 0.0, float(worst), float(dt_next)))
            if accept:
                # dopri5.py:39-45 + interp.py:21-35
                ymid = [y_ + _wsum(dts, DP_C_MID, k_) for y_, k_ in zip(y_cur, k)]
                f0 = [k_[0] for k_ in k]
                ca = [_dot([-2 * dts, 2 * dts, -8, -8, 16], [p, q, r_, s, m]) for p, q, r_, s, m in zip(f0, f1, y_cur, y1, ymid)]
                cb = [_dot([5 * dts, -3 * dts, 18, 14, -32], [p, q, r_, s, m]) for p, q, r_, s, m in zip(f0, f1, y_cur, y1, ymid)]
                cc = [_dot([-4 * dts, dts, -11, -5, 16], [p, q, r_, s, m]) for p, q, r_, s, m in zip(f0, f1, y_cur, y1, ymid)]
                cd = [dts * p for p in f0]
                coeff = (ca, cb, cc, cd, y_cur)
                y_cur, f_cur = y1, f1
                t_lo, t_hi = t0, t0 + dt
            else:
                t_lo = t_hi = t0
            dt = dt_next
        # --- interp.py:51-65 (x formed in the state dtype)
        a0, a1, at = t_lo.type(dtype), t_hi.type(dtype), nxt.type(dtype)
        assert bool((a0 <= at) & (at <= a1)), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(a0, at, a1)
        x = ((at - a0) / (a1 - a0)).type(dtype)
        xs = [torch.tensor(1).type(dtype), x]
        for _ in range(2, 5):
            xs.append(xs[-1] * x)
        sol.append([_dot(cs, reversed(xs)) for cs in zip(*coeff)])
    return sol


# ------------------------------------------------------------------------------------------------
# variable-coefficient Adams-Bashforth-Moulton (method 'adams')   torchdiffeq/_impl/adams.py:11-170
# ------------------------------------------------------------------------------------------------

# adams.py:11-15
GAMMA_STAR = [
    1, -1 / 2, -1 / 12, -1 / 24, -19 / 720, -3 / 160, -863 / 60480, -275 / 24192, -33953 / 3628800, -0.00789255,
    -0.00678585, -0.00592406, -0.00523669, -0.0046775, -0.00421495, -0.0038269
]


def _adams_g_and_explicit_phi(prev_t, next_t, implicit_phi, k):
    """adams.py:26-50: the g coefficients (float64) and the explicit phi of the divided-difference form."""
    curr_t = prev_t[0]
    dt = next_t - prev_t[0]
    g = torch.empty(k + 1).to(prev_t[0])
    explicit_phi = [implicit_phi[0]]
    beta = torch.tensor(1).to(prev_t[0])
    g[0] = 1
    c = 1 / torch.arange(1, k + 2).to(prev_t[0])
    for j in range(1, k):
        beta = (next_t - prev_t[j - 1]) / (curr_t - prev_t[j]) * beta
        cast = beta.to(implicit_phi[j][0])
        explicit_phi.append([p * cast for p in implicit_phi[j]])
        c = c[:-1] - c[1:] if j == 1 else c[:-1] - c[1:] * dt / (next_t - prev_t[j - 1])
        g[j] = c[0]
    c = c[:-1] - c[1:] * dt / (next_t - prev_t[k - 1])
    g[k] = c[0]
    return g, explicit_phi


def _adams_implicit_phi(explicit_phi, f_n, k):
    """adams.py:53-59."""
    k = min(len(explicit_phi) + 1, k)
    out = [f_n]
    for j in range(1, k):
        out.append([a - b for a, b in zip(out[j - 1], explicit_phi[j - 1])])
    return out


def _adams_step_size(last_step, ratios, safety, ifactor, dfactor, order):
    """misc.py:160-170 with the `order` argument adams passes (the exponent is born a float32 tensor)."""
    worst = max(ratios)
    if worst == 0:
        return last_step * ifactor
    if worst < 1:
        dfactor = torch.tensor(1, dtype=torch.float64)
    er = torch.sqrt(worst).to(last_step)
    expo = torch.tensor(1 / order).to(last_step)
    factor = torch.max(1 / ifactor, torch.min(er ** expo / safety, 1 / dfactor))
    return last_step / factor


def _adams(fn, y_init, t, rtol, atol, step_log, max_order=12, safety=0.9, ifactor=10.0, dfactor=0.2):
    """adams.py:62-170.  `step_log` rows: (t_n, attempted next_t, order, accepted, max error ratio, next next_t)."""
    n = len(y_init)
    rtol = list(rtol) if isinstance(rtol, (list, tuple)) else [rtol] * n
    atol = list(atol) if isinstance(atol, (list, tuple)) else [atol] * n
    max_order = int(max(1, min(max_order, 12)))
    safety, ifactor, dfactor = ((v if torch.is_tensor(v) else torch.tensor(v)).type(torch.float64) for v in (safety, ifactor, dfactor))
    t = t.to(torch.float64)                                                    # solvers.py:28
    # before_integrate (:82-94)
    f0 = fn(t[0].type_as(y_init[0]), y_init)
    prev_t, prev_f, phi = [t[0]], [f0], [f0]                                    # index 0 = newest (deque.appendleft)
    first_step = _initial_step(fn, t[0], y_init, 2, rtol[0], atol[0], f0).to(t)
    y_n, next_t, order = y_init, t[0] + first_step, 1
    sol = [y_init]

    def ratio(coef, phis, tol):                                                 # misc.py:146-157 on dt * coef * phi
        out = []
        for p, tl in zip(phis, tol):
            r = (coef * p) / tl
            out.append(torch.mean(r * r))
        return out

    for i in range(1, len(t)):
        final_t = t[i]
        while final_t > prev_t[0]:                                              # advance (:96-101)
            y0 = y_n
            nt = final_t if next_t > final_t else next_t
            dt = nt - prev_t[0]
            dt_cast = dt.to(y0[0])
            g, ephi = _adams_g_and_explicit_phi(prev_t, nt, phi, order)
            g = g.to(y0[0])
            m = max(1, order - 1)
            p_next = [y_ + _wsum(dt_cast, g[:m], [ephi[j][q] for j in range(m)]) for q, y_ in enumerate(y0)]
            nf = fn(nt.to(p_next[0]), p_next)
            iphi_p = _adams_implicit_phi(ephi, nf, order + 1)
            y_next = [p_ + dt_cast * g[order - 1] * ip for p_, ip in zip(p_next, iphi_p[order - 1])]
            tol = [a_ + r_ * torch.max(torch.abs(u), torch.abs(v)) for a_, r_, u, v in zip(atol, rtol, y0, y_next)]
            error_k = ratio(dt_cast * (g[order] - g[order - 1]), iphi_p[order], tol)
            accept = bool((torch.tensor(error_k) <= 1).all())
            if not accept:
                dt_next = _adams_step_size(dt, error_k, safety, ifactor, dfactor, order)
                if step_log is not None:
                    step_log.append((float(prev_t[0]), float(nt), order, 0.0, float(max(error_k)), float(prev_t[0] + dt_next)))
                next_t = prev_t[0] + dt_next
                continue
            nf = fn(nt.to(p_next[0]), y_next)
            iphi = _adams_implicit_phi(ephi, nf, order + 2)
            next_order = order
            if len(prev_t) <= 4 or order < 3:
                next_order = min(order + 1, 3, max_order)
            else:
                e1 = ratio(dt_cast * (g[order - 1] - g[order - 2]), iphi_p[order - 1], tol)
                e2 = ratio(dt_cast * (g[order - 2] - g[order - 3]), iphi_p[order - 2], tol)
                if min(e1 + e2) < max(error_k):
                    next_order = order - 1
                elif order < max_order:
                    e3 = ratio(dt_cast * GAMMA_STAR[order], iphi_p[order], tol)
                    if max(e3) < max(error_k):
                        next_order = order + 1
            dt_next = dt if next_order > order else _adams_step_size(dt, error_k, safety, ifactor, dfactor, order + 1)
            if step_log is not None:
                step_log.append((float(prev_t[0]), float(nt), order, 1.0, float(max(error_k)), float(nt + dt_next)))
            prev_f = ([nf] + prev_f)[:max_order + 1]
            prev_t = ([nt] + prev_t)[:max_order + 1]
            y_n, next_t, phi, order = p_next, nt + dt_next, iphi[:max_order], next_order      # NB the PREDICTOR value is kept (:170)
        assert final_t == prev_t[0]
        sol.append(y_n)
    return sol


# ------------------------------------------------------------------------------------------------
# model-level wrappers (NDCN, dgnn block)
# ------------------------------------------------------------------------------------------------


def ndcn_forward(sd, A, t, x, method, rtol=.01, atol=.001, no_embed=False, no_graph=False, no_control=False):
    """NDCN.forward (neural_dynamics.py:150-160) from a state_dict `sd` keyed like the reference's."""
    if not no_embed:
        x = F.linear(torch.tanh(F.linear(x, sd['input_layer.0.weight'], sd['input_layer.0.bias'])),
                     sd['input_layer.2.weight'], sd['input_layer.2.bias'])
    f = OracleODEFunc(A, sd['neural_dynamic_layer.odefunc.wt.weight'], sd['neural_dynamic_layer.odefunc.wt.bias'],
                      no_graph=no_graph, no_control=no_control)
    h = odeint(f, x, t.type_as(x), rtol=rtol, atol=atol, method=method)
    return F.linear(h, sd['output_layer.weight'], sd['output_layer.bias'])


# ------------------------------------------------------------------------------------------------
# graph / operator builders (dense, as the reference builds them; small N only)
# ------------------------------------------------------------------------------------------------


def grid_8_neighbor_dense(S):
    """utils_in_learn_dynamics.py:137-157: node (x, y) -> x*S + y, 8 neighbours, no wrap-around."""
    S = int(S)
    A = np.zeros((S * S, S * S), dtype=np.float32)
    for x in range(S):
        for y in range(S):
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if (dx or dy) and 0 <= x + dx < S and 0 <= y + dy < S:
                        A[x * S + y, (x + dx) * S + (y + dy)] = 1
    return A


def _inv_sqrt_deg(deg):
    # the reference's np.power(deg, -0.5, where=deg != 0) leaves zero-degree slots undefined
    # (SURVEY A10 latent bug); defined as 0 here and in the build
    deg = np.asarray(deg, dtype=np.float32)
    out = np.zeros_like(deg)
    nz = deg != 0
    out[nz] = np.power(deg[nz], -0.5)
    return out


def normalized_laplacian_dense(A):
    """utils_in_learn_dynamics.py:109-120."""
    do, di = _inv_sqrt_deg(A.sum(1)), _inv_sqrt_deg(A.sum(0))
    return np.eye(A.shape[0]) - np.diag(do) @ A @ np.diag(di)


def normalized_adj_dense(A):
    """utils_in_learn_dynamics.py:123-134."""
    do, di = _inv_sqrt_deg(A.sum(1)), _inv_sqrt_deg(A.sum(0))
    return np.diag(do) @ A @ np.diag(di)


def zipf_smoothing_dense(A):
    """utils_in_learn_dynamics.py:80-92."""
    Ap = A + np.eye(A.shape[0])
    do, di = _inv_sqrt_deg(Ap.sum(1)), _inv_sqrt_deg(Ap.sum(0))
    return np.diag(do) @ Ap @ np.diag(di)


def laplacian_dense(A):
    """heat_dynamics.py:116-117."""
    return np.diag(A.sum(1)) - A


def zipf_smoothing_alpha(adj, alpha):
    """propagation.py:91-103 on a scipy CSR adjacency."""
    Ap = alpha * sp.eye(adj.shape[0]) + (1 - alpha) * adj
    do = _inv_sqrt_deg(np.asarray(Ap.sum(1)).reshape(-1))
    di = _inv_sqrt_deg(np.asarray(Ap.sum(0)).reshape(-1))
    return sp.csr_matrix(sp.diags(do) @ Ap @ sp.diags(di))


def x0_blocks(S):
    """heat_dynamics.py:178-182: three constant blocks 25 / 20 / 17 on the S x S canvas."""
    x0 = np.zeros((S, S), dtype=np.float32)
    x0[int(0.05 * S):int(0.25 * S), int(0.05 * S):int(0.25 * S)] = 25
    x0[int(0.45 * S):int(0.75 * S), int(0.45 * S):int(0.75 * S)] = 20
    x0[int(0.05 * S):int(0.25 * S), int(0.35 * S):int(0.65 * S)] = 17
    return x0.reshape(-1, 1)


# ------------------------------------------------------------------------------------------------
# known answers independent of the reference
# ------------------------------------------------------------------------------------------------


def heat_closed_form(L_dense, x0, t):
    """x(t) = V exp(-Lambda t) V^T x0 for symmetric L (the check back_up/heat_on_grid_old.py:94-100 points to)."""
    lam, V = np.linalg.eigh(np.asarray(L_dense, dtype=np.float64))
    c = V.T @ np.asarray(x0, dtype=np.float64)
    return np.stack([V @ (np.exp(-lam * float(ti))[:, None] * c) for ti in np.asarray(t, dtype=np.float64)])


def spmm_f64(indptr, indices, data, X):
    """Independent float64 CSR SpMM (scipy) for known-answer checks of the SpMM kernel."""
    m = sp.csr_matrix((np.asarray(data, dtype=np.float64), np.asarray(indices), np.asarray(indptr)),
                      shape=(len(indptr) - 1, X.shape[0]))
    return m @ np.asarray(X, dtype=np.float64)
