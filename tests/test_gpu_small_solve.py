"""The whole fixed-grid solve in one launch (csrc/solve_small.hip: ndcn_solve_small_f32 / _bwd_f32) for states that fit one
compute unit - the sizes of the reference's own commands (heat_dynamics.py:20-22,33: Euler, 400 nodes, H = 20).

Forward: every tick bit-identical to the per-step kernels (rhs_small + fixed_stage) and, through them, within the
fixtures' bounds of the reference.  Backward (Euler): gradients against the per-step autograd path and the CPU oracle's
autograd through the restated solver."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _case(dev, S=20, H=20, no_control=False, no_graph=False, seed=0, rect=None):
    from ndcn_amd import graphs
    from ndcn_amd.neural_dynamics import ODEFunc
    G = graphs.grid_8_neighbor(S) if rect is None else graphs.grid_8_neighbor_rect(*rect)
    L = graphs.normalized_laplacian(G)
    torch.manual_seed(seed)
    f = ODEFunc(H, graphs.to_device(L, dev), no_control=no_control, no_graph=no_graph).to(dev)
    x0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(seed + 1))
    return f, L, x0


@pytest.mark.parametrize('method', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('shape', [(20, 20), (24, 16), (7, 64), (20, 1), (9, 33)])
@pytest.mark.parametrize('variant', ['default', 'no_control', 'no_graph'])
def test_one_launch_solve_is_bit_identical_to_the_per_step_kernels(dev, method, shape, variant):
    """odeint on the device-resident solver takes the one-launch path (asserted through ndcn_solve_small_supported); the
    same solve stepped tick by tick (ndcn_solver_advance: one RHS launch + stage kernels per step) and through the generic
    host loop gives the same bits at every tick; irregular grid (heat_dynamics.py:129-147)."""
    from ndcn_amd import _lib
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.csr import as_csr
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    S, H = shape
    f, L, x0 = _case(dev, S, H, no_control=variant == 'no_control', no_graph=variant == 'no_graph')
    f.eval()
    t = torch.sort(torch.rand(37, generator=torch.Generator().manual_seed(5)) * 3.0).values
    t[0] = 0.0
    flags = _lib.F_RELU | (_lib.F_NO_GRAPH if f.no_graph else 0) | (_lib.F_NO_CONTROL if f.no_control else 0)
    assert _lib.load().ndcn_solve_small_supported(as_csr(f.A).view_ref(), H, flags, _lib.METHODS[method], 0) == 1
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), method=method)
        yg = ode.odeint(lambda tt, yy: f(tt, yy), x0.to(dev), t.to(dev), method=method)
        solver = DeviceSolver(f, x0.shape[0], method)
        ys = torch.empty_like(y)
        ys[0] = x0.to(dev)
        tt = t.to(torch.float32).to(torch.float64).tolist()
        solver.begin(ys[0], tt[0])
        for i in range(1, len(tt)):
            solver.advance(tt[i], ys[i])
        torch.cuda.synchronize()
        solver.close()
    assert torch.equal(y, ys)
    assert torch.equal(y, yg)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize('name', ['fixed_euler_equal', 'fixed_euler_irregular', 'fixed_midpoint_equal', 'fixed_midpoint_irregular',
                                  'fixed_rk4_equal', 'fixed_rk4_irregular'])
def test_one_launch_solve_against_the_reference_fixtures(dev, name):
    """The reference's own fixed-grid trajectories (fixtures G2: NDCN-style ODEFunc on the 400-node grid, H = 20, 100 ticks)."""
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden(name)
    n = int(d['x0'].shape[0])
    A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], (n, n), dev)
    f = ODEFunc(d['x0'].shape[1], A).to(dev).eval()
    f.load_state_dict({'wt.weight': torch.from_numpy(d['W']), 'wt.bias': torch.from_numpy(d['b'])})
    with torch.no_grad():
        y = ode.odeint(f, torch.from_numpy(d['x0']).to(dev), torch.from_numpy(d['t']).to(dev), method=name.split('_')[1])
    err = np.abs(y.cpu().numpy() - d['traj'])
    assert err.mean() < 1e-5 and err.max() < 2e-4


def test_more_ticks_than_one_launch_holds_and_sizes_out_of_range(dev):
    """300 ticks = three launches chained through the output panels (bit-identical to the per-step path); a state beyond
    one CU's LDS or register budget is declined and takes the per-step path."""
    from ndcn_amd import _lib, graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.csr import as_csr
    f, L, x0 = _case(dev, 20, 20)
    f.eval()
    t = torch.linspace(0., 4., 301)
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), method='euler')
        yg = ode.odeint(lambda tt, yy: f(tt, yy), x0.to(dev), t.to(dev), method='euler')
    assert torch.equal(y, yg)
    lib = _lib.load()
    big, _, xb = _case(dev, 40, 20)                                   # 1600 rows x 20: more passes than the registers hold
    assert lib.ndcn_solve_small_supported(as_csr(big.A).view_ref(), 20, _lib.F_RELU, _lib.M_EULER, 0) == 0
    wide, _, _ = _case(dev, 8, 128)
    assert lib.ndcn_solve_small_supported(as_csr(wide.A).view_ref(), 128, _lib.F_RELU, _lib.M_EULER, 0) == 0
    assert lib.ndcn_solve_small_supported(as_csr(f.A).view_ref(), 20, _lib.F_RELU, _lib.M_DOPRI5, 0) == 0
    assert lib.ndcn_solve_small_supported(as_csr(f.A).view_ref(), 20, _lib.F_RELU, _lib.M_RK4, 1) == 0      # backward: Euler only
    big.eval()
    with torch.no_grad():
        yb = ode.odeint(big, xb.to(dev), t[:5].to(dev), method='euler')
    assert torch.isfinite(yb).all()


@pytest.mark.parametrize('variant', ['default', 'no_control', 'no_graph'])
@pytest.mark.parametrize('shape', [(20, 20), (11, 31), (16, 8)])
def test_one_launch_euler_backward_equals_the_per_step_autograd_and_the_oracle(dev, variant, shape):
    """loss = sum_ticks <w_i, y(t_i)> through the README-sized Euler solve: gradients wrt y0, W, b from ONE reverse launch
    against (i) the per-step autograd path of this package (NDCN_SOLVE_SMALL_GRAD=0) and (ii) torch autograd through the
    oracle's restated solver on the CPU."""
    from ndcn_amd import torchdiffeq as ode
    S, H = shape
    f, L, x0 = _case(dev, S, H, no_control=variant == 'no_control', no_graph=variant == 'no_graph', seed=3)
    t = torch.sort(torch.rand(41, generator=torch.Generator().manual_seed(6)) * 2.0).values
    t[0] = 0.0
    wts = torch.randn(41, x0.shape[0], H, generator=torch.Generator().manual_seed(7))

    def run(flag):
        os.environ['NDCN_SOLVE_SMALL_GRAD'] = flag
        try:
            f.zero_grad()
            y0 = x0.to(dev).requires_grad_(True)
            y = ode.odeint(f, y0, t.to(dev), method='euler')
            (y * wts.to(dev)).sum().backward()
            return (y.detach().cpu(), y0.grad.cpu(), None if f.wt.weight.grad is None else f.wt.weight.grad.cpu().clone(),
                    None if f.wt.bias.grad is None else f.wt.bias.grad.cpu().clone())
        finally:
            del os.environ['NDCN_SOLVE_SMALL_GRAD']
    ya, gya, gWa, gba = run('1')
    yb, gyb, gWb, gbb = run('0')
    assert torch.equal(ya, yb)
    # oracle: autograd through the restated Euler loop
    A = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
    Wo = f.wt.weight.detach().cpu().clone().requires_grad_(True)
    bo = f.wt.bias.detach().cpu().clone().requires_grad_(True)
    yo0 = x0.clone().requires_grad_(True)
    fo = orc.OracleODEFunc(A, Wo, bo, no_graph=variant == 'no_graph', no_control=variant == 'no_control')
    yo = orc.odeint(fo, yo0, t, method='euler')
    (yo * wts).sum().backward()

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 2e-4 * scale, (what, float((a - b).abs().max()), scale)
    close(gya, gyb, 'g_y0 vs per-step')
    close(gya, yo0.grad, 'g_y0 vs oracle')
    if variant != 'no_control':
        close(gWa, gWb, 'g_W vs per-step')
        close(gba, gbb, 'g_b vs per-step')
        close(gWa, Wo.grad, 'g_W vs oracle')
        close(gba, bo.grad, 'g_b vs oracle')
    else:
        assert gWa is None or float(gWa.abs().max()) == 0.0


def test_readme_training_step_runs_on_two_launches(dev, capsys):
    """heat_dynamics.py's README command (grid, H = 20, Euler, equal sampling) through the driver counterpart: the ODE
    block's forward and backward are one launch each, and the loss still goes down."""
    from ndcn_amd.drivers.dynamics import main
    out = main('heat', ['--network', 'grid', '--sampled_time', 'equal', '--baseline', 'ndcn', '--gpu', '0',
                        '--niters', '40', '--test_freq', '10', '--time_tick', '20', '--method', 'euler'])
    text = capsys.readouterr().out
    lines = [l for l in text.splitlines() if l.startswith('Iter ')]
    first = float(lines[0].split('Train Loss ')[1].split('(')[0])
    last = float(lines[-1].split('Train Loss ')[1].split('(')[0])
    assert last < first


@pytest.mark.parametrize('method', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('variant', ['default', 'no_control', 'no_graph'])
@pytest.mark.parametrize('shape', [(45, 20), (30, 256), (12, 33)])
def test_fixed_grid_training_through_fused_launches_equals_the_per_op_autograd_path(dev, method, variant, shape):
    """Fixed-grid solves too large for one compute unit (or not Euler) train through `_FixedGridSolve`: the solver's fused
    launches forward, a closed-form reverse sweep backward (stages re-formed per step, SpMM / masked Linear backward / SpMM with
    A^T, the step size folded in).  Against the per-operation autograd path (NDCN_FIXED_GRID_GRAD=0: one autograd node per
    kernel): the same trajectory bit for bit, gradients wrt y0, W, b to summation-order rounding; H = 256 runs rhs_fused3 with
    its RK4 / COMBINE epilogues under grad."""
    from ndcn_amd import torchdiffeq as ode
    S, H = shape
    f, L, x0 = _case(dev, S, H, no_control=variant == 'no_control', no_graph=variant == 'no_graph', seed=11)
    t = torch.sort(torch.rand(9, generator=torch.Generator().manual_seed(6)) * 1.5).values
    t[0] = 0.0
    wts = torch.randn(9, x0.shape[0], H, generator=torch.Generator().manual_seed(7))

    def run(flags):
        os.environ.update(flags)
        try:
            f.zero_grad()
            y0 = x0.to(dev).requires_grad_(True)
            y = ode.odeint(f, y0, t.to(dev), method=method)
            (y * wts.to(dev)).sum().backward()
            return (y.detach().cpu(), y0.grad.cpu(), None if f.wt.weight.grad is None else f.wt.weight.grad.cpu().clone(),
                    None if f.wt.bias.grad is None else f.wt.bias.grad.cpu().clone())
        finally:
            for k in flags:
                del os.environ[k]
    ya, gya, gWa, gba = run({'NDCN_SOLVE_SMALL_GRAD': '0'})                          # (keep the one-launch path out of the way)
    yb, gyb, gWb, gbb = run({'NDCN_SOLVE_SMALL_GRAD': '0', 'NDCN_FIXED_GRID_GRAD': '0'})
    assert torch.equal(ya, yb)

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-4 * scale, (what, float((a - b).abs().max()), scale)
    close(gya, gyb, 'g_y0')
    if variant != 'no_control':
        close(gWa, gWb, 'g_W')
        close(gba, gbb, 'g_b')
    else:
        assert gWa is None or float(gWa.abs().max()) == 0.0


@pytest.mark.parametrize('side,H,n_ticks', [(20, 20, 80), (17, 16, 150), (20, 20, 150), (12, 16, 20)])
def test_euler_reverse_sweep_on_kept_intermediates_is_bit_identical(dev, side, H, n_ticks):
    """ndcn_solve_small_keep_f32 / _bwd_keep_f32: the forward launch keeps S_i = A y_i and K_i of every step, the sweep reads them
    instead of re-forming them - the same fma chains produced the kept values, so trajectory and gradients equal the recomputing pair's
    bit for bit (NDCN_SOLVE_SMALL_KEEP=0).  150 ticks: two launches per direction (128 ticks each at most)."""
    import os
    from ndcn_amd import graphs, _lib
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    n = side * side
    ticks = torch.linspace(0., 3.0, n_ticks + 1)
    x0h = torch.rand(n, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(n_ticks + 1, n, H, generator=torch.Generator().manual_seed(1))
    A = graphs.to_device(op, dev)
    # (12 x 12 x 16: too few elements for the fast sweep's row groups - nothing is kept, the pair must still agree)
    assert _lib.load().ndcn_solve_small_keep_supported(A.view_ref(need_symmetric=True), H, _lib.F_RELU) == (0 if side == 12 else 1)
    res = {}
    for keep in ('1', '0'):
        os.environ['NDCN_SOLVE_SMALL_KEEP'] = keep
        try:
            torch.manual_seed(0)
            f = ODEFunc(H, A).to(dev)
            x0 = x0h.clone().to(dev).requires_grad_(True)
            y = ode.odeint(f, x0, ticks.to(dev), method='euler')
            assert type(y.grad_fn).__name__ == '_SmallEulerSolveBackward'
            (y * w.to(dev)).sum().backward()
            res[keep] = (y.detach().cpu(), x0.grad.cpu(), f.wt.weight.grad.cpu(), f.wt.bias.grad.cpu())
        finally:
            del os.environ['NDCN_SOLVE_SMALL_KEEP']
    for a, b in zip(res['1'], res['0']):
        assert torch.equal(a, b)
