"""CPU tests of the product's HOST logic (ndcn_amd/torchdiffeq/_impl/core.py): the solver control flow is
driven with the oracle-backed ops double and compared with the fixtures captured from the reference.
(The HIP kernels themselves are tested on the GPU in test_gpu_*.py.)"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden
from oracle import ndcn_oracle as orc
from ndcn_amd.torchdiffeq._impl import core
from _oracle_ops import OracleOps

torch.set_num_threads(1)


def T(a):
    return torch.from_numpy(np.asarray(a))


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def run(func, y0, t, method, rtol=1e-7, atol=1e-9, log=None, **options):
    tensor_input, f, y, tt = core.check_inputs(func, y0, t)
    if method == 'dopri5':
        sol = core.integrate_dopri5(OracleOps, f, y, tt, rtol, atol, step_log=log, **options)
    elif method == 'adams':
        sol = core.integrate_adams(OracleOps, f, y, tt, rtol, atol, step_log=log, **options)
    else:
        sol = core.integrate_fixed(OracleOps, f, y, tt, method)
    out = tuple(torch.stack([s[i] for s in sol]) for i in range(len(y)))
    return out[0] if tensor_input else out


def make_func(d, **kw):
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    return orc.OracleODEFunc(A, T(d['W']), T(d['b']), **kw)


@pytest.mark.parametrize('name', names('fixed_*.npz'))
def test_fixed_grid_control_flow(name):
    d = load_golden(name)
    f = make_func(d)
    y = run(f, T(d['x0']), T(d['t']), name.split('_')[1])
    assert np.abs(y.numpy() - d['traj']).max() <= 1e-6


@pytest.mark.parametrize('name', names('adams_*.npz'))
def test_adams_control_flow(name):
    """core.Adams (variable-coefficient Adams-Bashforth-Moulton, adams.py:62-170) against the reference's own run: same
    attempted steps (t_n, next_t, order, accept / reject, following next_t), same number of evaluations, same states."""
    d = load_golden(name)
    f = make_func(d, no_control=bool(d['no_control']))
    log = []
    opts = {k[4:]: (int(v) if k == 'opt_max_order' else float(v)) for k, v in d.items() if k.startswith('opt_')}
    y = run(f, T(d['x0']), T(d['t']), 'adams', float(d['rtol']), float(d['atol']), log, **opts)
    nfe = dict([log.pop()])['nfe']
    ref, got = d['steplog'], np.array(log)
    assert got.shape[0] == ref.shape[0] and nfe == int(d['nfe'])
    assert np.array_equal(got[:, 2:4], ref[:, 2:4])                          # order and accept / reject, step by step
    # (step ends proposed from error ratios of ~1e-9 - pure cancellation noise - move by 1e-4 relative with the 1e-7 difference of
    # the initial step's float32 norms; accepted tick-clipped ends agree far better)
    assert np.allclose(got[:, :2], ref[:, :2], rtol=1e-3, atol=1e-9) and np.allclose(got[:, 5], ref[:, 4], rtol=2e-3)
    assert np.abs(y.numpy() - d['traj']).max() <= 1e-5 * max(1.0, np.abs(d['traj']).max())


@pytest.mark.parametrize('name', names('dopri5_*.npz'))
def test_dopri5_control_flow(name):
    d = load_golden(name)
    f = make_func(d, no_control='no_control' in name)
    log = []
    opts = {k[4:]: float(v) for k, v in d.items() if k.startswith('opt_')}
    y = run(f, T(d['x0']), T(d['t']), 'dopri5', float(d['rtol']), float(d['atol']), log, **opts)
    nfe = dict([log.pop()])['nfe']
    ref = d['steplog']
    log = np.array(log)
    scale = max(1.0, np.abs(d['traj']).max())
    # The host logic forms every scalar the way torch does on 0-d tensors (python_scalar / tensor = reciprocal * scalar,
    # tensor ** python_float in double, float32 means and norms in ATen's summation order - the ops double returns those):
    # with the oracle's panel arithmetic underneath, the reference's run is reproduced to the last bit - every row of the
    # per-attempt log and the trajectory, the rtol 1e-7 solve (dopri5_tight: 48 attempts, rejections) included.
    assert nfe == int(d['nfe'])
    assert log.shape == ref.shape and np.array_equal(log, ref)
    assert np.array_equal(y.numpy(), d['traj'])


def test_tuple_state_and_time_dependent_func():
    # tuple state + a func that really uses t: product host logic vs the oracle's own tuple path
    def f(t, y):
        a, b = y
        return (-a * t + b.mean(), torch.sin(t) * b - a.sum() * 0.01)
    g = torch.Generator().manual_seed(3)
    y0 = (torch.rand(7, 3, generator=g), torch.rand(5, generator=g))
    t = torch.linspace(0., 2., 9)
    for method, tol in (('euler', 2e-6), ('midpoint', 2e-6), ('rk4', 2e-6), ('dopri5', 1e-4)):
        ref = orc.odeint(f, y0, t, rtol=1e-4, atol=1e-6, method=method)
        got = run(f, y0, t, method, 1e-4, 1e-6)
        assert isinstance(got, tuple) and len(got) == 2
        for g, r in zip(got, ref):
            assert g.shape == r.shape
            assert (g - r).abs().max() <= tol


def test_rejected_steps_and_error_paths():
    # a discontinuous forcing term forces rejections
    f = lambda t, y: 100.0 * (t > 0.35).to(y.dtype) - y + torch.cos(3 * t)
    y0 = torch.zeros(4)
    t = torch.linspace(0., 1., 5)
    log_ref, log = [], []
    ref = orc.odeint(f, y0, t, rtol=1e-5, atol=1e-7, method='dopri5', step_log=log_ref)
    got = run(f, y0, t, 'dopri5', 1e-5, 1e-7, log)
    log.pop()
    assert any(r[2] == 0 for r in log_ref)
    assert [r[2] for r in log] == [r[2] for r in log_ref]
    assert (got - ref).abs().max() < 1e-5
    # non-finite state -> AssertionError (dopri5.py:101-102 / :100)
    with pytest.raises(AssertionError):
        run(lambda t, y: y * float('inf'), torch.ones(3), torch.tensor([0., 1.]), 'dopri5', 1e-3, 1e-3)
    with pytest.raises(AssertionError):
        run(f, y0, torch.tensor([0., 1., 0.5]), 'dopri5')
    with pytest.raises(TypeError):
        core.check_inputs(f, torch.ones(3, dtype=torch.int32), torch.tensor([0., 1.]))
    with pytest.raises(TypeError):
        core.check_inputs(f, y0, torch.tensor([0, 1]))


def test_step_size_controller_matches_reference_formula():
    # misc.py:160-170 incl. the float32-born constants; fixtures give (dt, ratio) -> dt_next
    for name in names('dopri5_*.npz'):
        d = load_golden(name)
        k = {n[4:]: core.controller_constant(float(v)) for n, v in d.items() if n.startswith('opt_')}
        for t0, dt, acc, ratio, dt_next in d['steplog']:
            got = core.optimal_step_size(dt, np.float32(ratio), **k)
            assert abs(got - dt_next) <= 1e-6 * abs(dt_next), (name, dt, ratio)
    d = load_golden('dopri5_options')                       # non-default safety / ifactor / dfactor, every branch
    k = {n: core.controller_constant(float(d['opt_' + n])) for n in ('safety', 'ifactor', 'dfactor')}
    for (dt, ratio), want in zip(d['ctl_in'], d['ctl_out']):
        got = core.optimal_step_size(dt, np.float32(ratio), **k)
        assert abs(got - want) <= 1e-12 * abs(want), (dt, ratio, got, want)
    assert core.optimal_step_size(0.5, np.float32(0)) == 5.0
    assert np.isnan(core.optimal_step_size(0.5, np.float32('nan')))


def test_bench_cpu_legs_on_a_tiny_sample(monkeypatch):
    """bench.py's CPU legs (oracle only, no device): the at-scale point - one timed right-hand side and one one-step solve,
    extrapolated - and the fixed-grid sample of `--method euler`, at sizes that take milliseconds."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    monkeypatch.setitem(bench.AT_SCALE_NODES, 'M', 12 * 12)
    monkeypatch.setitem(bench.AT_SCALE_NODES, 'C2', 300)
    r = bench.cpu_at_scale('M', 8, 5.0, .01, .001)
    assert r['nodes'] == 144 and r['steps_in_it'] == 1 and r['rhs_evals_in_it'] == 8 and r['value_extrapolated'] > 0
    r2 = bench.cpu_at_scale('C2', 8, 5.0, .01, .001)
    assert r2['rhs_evals_in_it'] == 4 and r2['steps_in_it'] == 1                      # one RK4 step
    assert bench.cpu_at_scale('C5', 8, 5.0, .01, .001) is None                         # C5's CPU leg already runs the full size
    monkeypatch.setitem(bench.CPU_SAMPLE, 'M', 8 * 8)
    monkeypatch.setattr(bench, 'FIXED_GRID_METHOD', 'euler')
    base, parity = bench.cpu_baseline('M', 8, 5.0, .01, .001, threads=2, runs=5, dev=None, at_scale=False)
    assert parity is None and base['kind'] == 'port' and base['value'] > 0 and 'euler' in base['sample'] and 'at_scale' not in base


def test_split_product_scale_target_emulation():
    """csrc/split16.h's guarantee, emulated on the CPU (tools/micro/split_emul.py: fp16 pieces incl. subnormals, three products, exact
    accumulation): with the row maximum at [2^14, 2^15) and one scale per output row of W every outlier family of the round-4 review
    stays below 1e-6 of sum |s w|; with the old target [0.5, 1) and one global weight scale the same operands lose up to 10 bits."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('split_emul', os.path.join(ROOT, 'tools', 'micro', 'split_emul.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(0)
    n, H = 128, 256
    S = rng.random((n, H))
    W = (rng.random((H, H)) - 0.5) / 8

    def ratio(S_, W_, top, per_row):
        S_, W_ = S_.astype(np.float32), W_.astype(np.float32)
        ref = S_.astype(np.float64) @ W_.astype(np.float64).T
        mag = np.abs(S_).astype(np.float64) @ np.abs(W_).astype(np.float64).T
        return float(np.max(np.abs(m.split_product(S_, W_, top, per_row) - ref) / mag))

    W12 = W.copy(); W12[17, 33] *= 2.0 ** 12
    Wrow = W.copy(); Wrow[17, :] *= 2.0 ** 12
    Sch = S.copy(); Sch[:, 5] *= 2.0 ** 12
    Wch = W.copy(); Wch[:, 5] *= 2.0 ** -12
    Wlog = 10.0 ** rng.uniform(-6, 0, size=(H, H)) * np.sign(rng.random((H, H)) - 0.5)
    for S_, W_ in ((S, W), (S, W12), (S, Wrow), (Sch, Wch), (Sch, W), (S, Wlog)):
        assert ratio(S_, W_, 15, True) < 1e-6
    assert ratio(S, W12, 0, False) > 1e-5 and ratio(S, Wrow, 0, False) > 1e-5 and ratio(Sch, Wch, 0, True) > 1e-5


def test_step_coefficient_table_is_bit_identical_to_the_products():
    """autograd_path._step_coefficients: dt * beta_ij and dt * c_err_j as ONE product with the tableau (so that `dts` has one
    consumer: deterministic gradient accumulation) must give the very bits of the 28 separate float32 products."""
    from ndcn_amd.torchdiffeq._impl import autograd_path as ap, core
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        d = (torch.rand((), generator=g) * 10.0 ** float(torch.randint(-6, 2, (), generator=g))).to(torch.float32)
        rows, cerr = ap._step_coefficients(d)
        for row, ref in zip(rows, core.DP_BETA):
            assert all(torch.equal(a, d * b) for a, b in zip(row, ref))
        assert all(torch.equal(a, d * c) for a, c in zip(cerr, core.DP_C_ERR))


def test_fan_out_adds_gradients_in_a_fixed_order():
    """autograd_path._FanOut: n aliases of one tensor, their gradients added left to right by one call (here: on the host) - the
    same sum whatever order the consumers' backward nodes run in."""
    from ndcn_amd.torchdiffeq._impl import autograd_path as ap
    x = torch.randn(64, dtype=torch.float32).requires_grad_(True)
    a, b, c, d = ap._fan(x, 4)
    (a * 1e8).sum().backward(retain_graph=True)
    assert torch.equal(x.grad, torch.full_like(x, 1e8))
    x.grad = None
    loss = (a * 3.0).sum() + (b * b).sum() + c.exp().sum() + (d * 1e-3).sum()
    loss.backward()
    xd = x.detach()
    want = ((torch.full_like(xd, 3.0) + 2 * xd) + xd.exp()) + torch.full_like(xd, 1e-3)      # left to right, float32
    assert torch.equal(x.grad, want)
    y = torch.randn(8)                                               # no gradient required: the tensor itself, n times
    assert all(t is y for t in ap._fan(y, 5))


def test_training_tape_scope_and_exact_sqrt(monkeypatch):
    """When does a dopri5 solve with gradient take the native tape (csrc/tape.hip)?  A plain ODEFunc state with a constant time grid;
    NDCN_GRAD_TAPE=0 / NDCN_VJP=torch / a time grid with gradient keep the per-operation graph.  And the controller chain's square
    root is the correctly rounded one (torch.sqrt is one ulp off where PyTorch dispatches AVX-512)."""
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq._impl import tape
    from ndcn_amd.torchdiffeq._impl.autograd_path import _Sqrt32
    f = ODEFunc(8, None, no_graph=True)
    y = torch.zeros(5, 8)
    t = torch.linspace(0, 1, 3)
    assert tape.applicable(f, y, t)
    assert not tape.applicable(f, y, t.clone().requires_grad_(True))
    assert not tape.applicable(f, y.double(), t)
    assert not tape.applicable(f, torch.zeros(5, 2, 8), t)
    monkeypatch.setenv('NDCN_GRAD_TAPE', '0')
    assert not tape.applicable(f, y, t)
    monkeypatch.delenv('NDCN_GRAD_TAPE')
    monkeypatch.setenv('NDCN_VJP', 'torch')
    assert not tape.applicable(f, y, t)
    for v in (5.235566646888401e-08, 0.3, 2.0, 1e-30):
        x = torch.tensor(np.float32(v), requires_grad=True)
        r = _Sqrt32.apply(x)
        assert float(r) == float(np.sqrt(np.float32(v)))
        r.backward()
        assert abs(float(x.grad) - 0.5 / float(np.sqrt(np.float32(v)))) <= 1e-6 * 0.5 / float(np.sqrt(np.float32(v)))
