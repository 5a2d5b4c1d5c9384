"""The native training tape (csrc/tape.hip: ndcn_tape_dopri5_f32 / ndcn_tape_backward_f32) against the per-operation autograd path
(autograd_path.integrate_dopri5_grad, itself checked against the reference's gradients and the oracle's autograd in
test_gpu_autograd.py): the SAME launches forward - trajectory and accept / reject log bit for bit - and the same gradient up to the
order of float32 sums, including the part that flows through the step-size controller (reference: torchdiffeq/_impl/dopri5.py:76-122,
misc.py:84-170, interp.py:38-65 differentiated by autograd)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need a ROCm device'
    return torch.device('cuda:0')


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def _solve(dev, tape, make_func, x0_host, ticks, rtol, atol, w_host):
    from ndcn_amd import torchdiffeq as ode
    os.environ['NDCN_GRAD_TAPE'] = '1' if tape else '0'
    try:
        f = make_func()
        x0 = x0_host.clone().to(dev).requires_grad_(True)
        log = []
        y = ode.odeint(f, x0, torch.tensor(ticks).to(dev), rtol=rtol, atol=atol, method='dopri5', step_log=log)
        (y * w_host.to(dev)).sum().backward()
        grads = [x0.grad.cpu()] + [p.grad.cpu() for p in f.parameters() if p.grad is not None]
        return y.detach().cpu(), log, grads
    finally:
        del os.environ['NDCN_GRAD_TAPE']


@pytest.mark.parametrize('variant', ['default', 'no_control', 'no_graph'])
@pytest.mark.parametrize('ticks,rtol,atol', [([0., 0.3, 0.6, 0.9, 1.0], 1e-3, 1e-5), (list(np.linspace(0., 5., 80)), 1e-2, 1e-3)])
def test_tape_equals_the_per_operation_path_on_the_reference_size(dev, variant, ticks, rtol, atol):
    """400 nodes x 20 hidden (heat_dynamics.py:33): narrow-panel kernels, ATen-order error norms, up to 7 ticks per dense pass"""
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden('fixed_rk4_equal')
    x0 = torch.from_numpy(np.asarray(d['x0'], dtype=np.float32))
    w = torch.randn(len(ticks), *x0.shape, generator=torch.Generator().manual_seed(3))

    def make():
        f = ODEFunc(20, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev), no_control=variant == 'no_control',
                    no_graph=variant == 'no_graph').to(dev)
        f.load_state_dict({'wt.weight': torch.from_numpy(np.asarray(d['W'], dtype=np.float32)),
                           'wt.bias': torch.from_numpy(np.asarray(d['b'], dtype=np.float32))})
        return f

    ya, la, ga = _solve(dev, True, make, x0, ticks, rtol, atol, w)
    yb, lb, gb = _solve(dev, False, make, x0, ticks, rtol, atol, w)
    assert la == lb and torch.equal(ya, yb)
    assert len(ga) == len(gb)
    for a, b in zip(ga, gb):
        assert rel(a, b) < 2e-4, rel(a, b)


@pytest.mark.parametrize('side,no_control', [(12, False), (36, False), (36, True)])
def test_tape_equals_the_per_operation_path_at_the_fused_width(dev, side, no_control):
    """H = 256: the fused MFMA launches with the stage algebra in their epilogues; side 36 (331 776 elements) is past the ATen-order
    range, so the error record rides in the seventh evaluation (both paths)."""
    from ndcn_amd import graphs
    from ndcn_amd.neural_dynamics import ODEFunc
    H = 256
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    ticks = [0., 0.4, 0.9, 1.5]
    x0 = torch.rand(side * side, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(4, side * side, H, generator=torch.Generator().manual_seed(1))

    def make():
        torch.manual_seed(0)
        return ODEFunc(H, graphs.to_device(op, dev), no_control=no_control).to(dev)

    ya, la, ga = _solve(dev, True, make, x0, ticks, 1e-3, 1e-4, w)
    yb, lb, gb = _solve(dev, False, make, x0, ticks, 1e-3, 1e-4, w)
    assert la == lb and torch.equal(ya, yb)
    assert len([r for r in la if r[0] != 'nfe']) >= 3
    for a, b in zip(ga, gb):
        assert rel(a, b) < 2e-4, rel(a, b)


def test_tape_with_rejected_attempts_and_a_power_law_graph(dev):
    """rejected attempts (their stage gradients flow through the ratio only), a hub-heavy operator, and the reference's error texts"""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    n, H = 1500, 32
    op = graphs.normalized_laplacian(graphs.barabasi_albert(n, 4, seed=1))
    ticks = [0., 0.01, 0.02, 0.9, 1.0, 2.5]
    x0 = 25.0 * torch.rand(n, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(len(ticks), n, H, generator=torch.Generator().manual_seed(1))

    def make():
        torch.manual_seed(0)
        return ODEFunc(H, graphs.to_device(op, dev)).to(dev)

    ya, la, ga = _solve(dev, True, make, x0, ticks, 1e-5, 1e-7, w)
    yb, lb, gb = _solve(dev, False, make, x0, ticks, 1e-5, 1e-7, w)
    assert la == lb and torch.equal(ya, yb)
    assert any(r[2] == 0.0 for r in la if r[0] != 'nfe')
    for a, b in zip(ga, gb):
        # (the gradient through a controller that rejects steps is ill-conditioned: test_gpu_autograd's carry-form test has the figures)
        assert rel(a, b) < 5e-2, rel(a, b)
    f = make()
    with pytest.raises(AssertionError, match='max_num_steps exceeded'):
        ode.odeint(f, x0.to(dev).requires_grad_(True), torch.tensor(ticks).to(dev), rtol=1e-5, atol=1e-7, method='dopri5',
                   options={'max_num_steps': 2})


def test_tape_without_polled_records(dev):
    """NDCN_POLL_RECORD=0: the reduction records come back by copy + event instead of through pinned host memory (csrc/hostrec.h);
    the switch is read once per process, so the case runs in a process of its own."""
    import subprocess
    import sys
    env = dict(os.environ, NDCN_POLL_RECORD='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.abspath(__file__), '-k',
                        'reference_size and default and ticks0 or fused_width and 36-False'], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


def test_second_backward_through_the_tape(dev):
    """retain_graph keeps the forward record (its device blocks are saved tensors of the node): the reverse pass runs again and gives the
    same gradient; once the graph is released, autograd's own error (round-5 advisor: the reference's graph is re-runnable)"""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(10))
    f = ODEFunc(16, graphs.to_device(op, dev)).to(dev)
    x0 = torch.rand(100, 16, device=dev, requires_grad=True)
    y = ode.odeint(f, x0, torch.tensor([0., 0.5, 1.0], device=dev), rtol=1e-3, atol=1e-4, method='dopri5')
    assert type(y.grad_fn).__name__.startswith('_TapeDopri5')
    y.sum().backward(retain_graph=True)
    g1, w1 = x0.grad.clone(), f.wt.weight.grad.clone()
    x0.grad = None
    f.zero_grad(set_to_none=True)
    # Jacobian rows: several grad calls over one solve
    rows = [torch.autograd.grad(y[2, i].sum(), x0, retain_graph=True)[0] for i in range(2)]
    assert not torch.equal(rows[0], rows[1])
    y.sum().backward()
    assert torch.equal(x0.grad, g1) and torch.equal(f.wt.weight.grad, w1)
    with pytest.raises(RuntimeError, match='second time'):
        y.sum().backward()


def test_tape_with_more_ticks_in_a_step_than_one_read_back_carries(dev):
    """a dense time grid over few accepted steps: > 24 dense-output groups (7 ticks each) inside one step - their inner products come
    back in batches (round-5 advisor: EINVAL 'too many dense-output groups' in backward only)"""
    from ndcn_amd import graphs
    from ndcn_amd.neural_dynamics import ODEFunc
    side, H = 10, 16
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    ticks = list(np.linspace(0., 2., 600))
    x0 = torch.rand(side * side, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(len(ticks), side * side, H, generator=torch.Generator().manual_seed(1))

    def make():
        torch.manual_seed(0)
        return ODEFunc(H, graphs.to_device(op, dev)).to(dev)

    ya, la, ga = _solve(dev, True, make, x0, ticks, 1e-2, 1e-3, w)
    yb, lb, gb = _solve(dev, False, make, x0, ticks, 1e-2, 1e-3, w)
    assert la == lb and torch.equal(ya, yb)
    steps = [r for r in la if r[0] != 'nfe']
    assert len(ticks) / max(len(steps), 1) > 7 * 24, (len(steps), 'the case must put > 24 groups into one step')
    for a, b in zip(ga, gb):
        assert rel(a, b) < 2e-4, rel(a, b)


def test_in_place_edit_of_a_fixed_grid_trajectory_before_backward_is_refused(dev):
    """the reverse pass re-forms the stages from the returned trajectory: it is saved through autograd (round-5 advisor)"""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    side, H = 24, 256                                                   # (beyond the one-launch solve: _NativeFixedGrid)
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    f = ODEFunc(H, graphs.to_device(op, dev)).to(dev)
    x0 = torch.rand(side * side, H, device=dev, requires_grad=True)
    y = ode.odeint(f, x0, torch.linspace(0., 1., 5, device=dev), method='rk4')
    assert type(y.grad_fn).__name__.startswith('_NativeFixedGrid')
    loss = (y * 1.0).sum()
    with torch.no_grad():
        y[2].mul_(2.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        loss.backward()


@pytest.mark.parametrize('method', ['euler', 'midpoint', 'rk4'])
@pytest.mark.parametrize('shape', ['reference_size', 'no_control', 'no_graph', 'fused_width'])
def test_native_fixed_grid_training_equals_the_python_loops(dev, method, shape):
    """ndcn_fixed_grid_train_f32 / _backward_f32 issue the launches of `_FixedGridSolve` (whose gradients test_gpu_autograd.py pins to
    the reference's): trajectory and the state's gradient bit for bit, the parameter gradients up to the rounding of their
    accumulation (torch's `add_(alpha=)` contracts to an fma, the library's combine rounds product and sum separately)."""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    side, H = (20, 20) if shape != 'fused_width' else (24, 256)
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    n = side * side
    ticks = torch.linspace(0., 1.5, 9)
    x0h = torch.rand(n, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(9, n, H, generator=torch.Generator().manual_seed(1))
    res = {}
    for native in ('1', '0'):
        os.environ['NDCN_FIXED_GRID_NATIVE'] = native
        os.environ['NDCN_SOLVE_SMALL_GRAD'] = '0'                      # (Euler at this size would take the one-launch kernel)
        try:
            torch.manual_seed(0)
            f = ODEFunc(H, graphs.to_device(op, dev), no_control=shape == 'no_control', no_graph=shape == 'no_graph').to(dev)
            x0 = x0h.clone().to(dev).requires_grad_(True)
            y = ode.odeint(f, x0, ticks.to(dev), method=method)
            (y * w.to(dev)).sum().backward()
            res[native] = (y.detach().cpu(), x0.grad.cpu(), [p.grad.cpu() for p in f.parameters() if p.grad is not None])
        finally:
            del os.environ['NDCN_FIXED_GRID_NATIVE'], os.environ['NDCN_SOLVE_SMALL_GRAD']
    assert torch.equal(res['1'][0], res['0'][0])
    assert torch.equal(res['1'][1], res['0'][1])
    assert len(res['1'][2]) == len(res['0'][2]) == (0 if shape == 'no_control' else 2)
    for a, b in zip(res['1'][2], res['0'][2]):
        assert rel(a, b) < 2e-6, rel(a, b)


def test_in_place_parameter_change_between_forward_and_backward_is_refused(dev):
    """the reverse passes read W again: autograd's saved-tensor version check must see the change (as for any torch op)"""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(10))
    for method in ('dopri5', 'rk4'):
        f = ODEFunc(16, graphs.to_device(op, dev)).to(dev)
        x0 = torch.rand(100, 16, device=dev, requires_grad=True)
        y = ode.odeint(f, x0, torch.tensor([0., 0.5, 1.0], device=dev), rtol=1e-3, atol=1e-4, method=method)
        with torch.no_grad():
            f.wt.weight.mul_(2.0)
        with pytest.raises(RuntimeError, match='modified by an inplace operation'):
            y.sum().backward()


@pytest.mark.parametrize('method', ['midpoint', 'rk4'])
@pytest.mark.parametrize('shape', ['reference_size', 'no_control', 'no_graph', 'four_passes', 'asymmetric'])
def test_one_launch_midpoint_and_rk4_reverse_sweeps(dev, method, shape):
    """solve_small.hip: the whole midpoint / RK4 solve and its reverse sweep in one launch each (states that fit one compute unit)
    against the library's multi-launch loops (NDCN_SOLVE_SMALL_RK_GRAD=0: ndcn_fixed_grid_*): the same trajectory bit for bit, the
    gradients up to the order of their float32 sums."""
    import scipy.sparse as sp
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    side, H = (20, 20) if shape != 'four_passes' else (10, 16)
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    if shape == 'asymmetric':
        op = sp.csr_matrix(sp.triu(op, k=-1) + 0.3 * sp.tril(op, k=-2))
    n = side * side
    ticks = torch.linspace(0., 2.0, 12)
    x0h = torch.rand(n, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(12, n, H, generator=torch.Generator().manual_seed(1))
    res = {}
    for one_launch in ('1', '0'):
        os.environ['NDCN_SOLVE_SMALL_RK_GRAD'] = one_launch
        os.environ['NDCN_SOLVE_SMALL_RK_MAX'] = '1000000'               # (read once per process: the default gate is 4 608 elements)
        try:
            torch.manual_seed(0)
            f = ODEFunc(H, graphs.to_device(op, dev), no_control=shape == 'no_control', no_graph=shape == 'no_graph').to(dev)
            x0 = x0h.clone().to(dev).requires_grad_(True)
            y = ode.odeint(f, x0, ticks.to(dev), method=method)
            (y * w.to(dev)).sum().backward()
            res[one_launch] = (y.detach().cpu(), x0.grad.cpu(), [p.grad.cpu() for p in f.parameters() if p.grad is not None])
            assert (type(y.grad_fn).__name__ == '_SmallEulerSolveBackward') == (one_launch == '1'), type(y.grad_fn).__name__
        finally:
            del os.environ['NDCN_SOLVE_SMALL_RK_GRAD'], os.environ['NDCN_SOLVE_SMALL_RK_MAX']
    assert torch.equal(res['1'][0], res['0'][0])
    assert rel(res['1'][1], res['0'][1]) < 1e-5, rel(res['1'][1], res['0'][1])
    assert len(res['1'][2]) == len(res['0'][2])
    for a, b in zip(res['1'][2], res['0'][2]):
        assert rel(a, b) < 1e-5, rel(a, b)
