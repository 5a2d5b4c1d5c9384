"""End-to-end parity of the drop-in API on a real MI355X: odeint / ODEBlock / NDCN / truth dynamics
against the golden fixtures captured from the reference and against the CPU oracle.
Tolerance: north_star's trajectory L1 < 1e-4 (tests use tighter max-abs bounds where the data allow)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu
L1_TOL = 1e-4          # BASELINE.json north_star: trajectory L1 error vs CPU reference < 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.asarray(a))


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def make_func(d, dev, as_module=True, **kw):
    """our ODEFunc (device-resident path) or a plain closure over hip ops (generic path)."""
    from ndcn_amd import CsrOperator, hip
    from ndcn_amd.neural_dynamics import ODEFunc
    A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)
    H = d['W'].shape[0]
    if as_module:
        f = ODEFunc(H, A, **kw).to(dev)
        f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
        return f.eval()
    W, b = T(d['W']).to(dev), T(d['b']).to(dev)
    return lambda t, x: hip.rhs(A, x, W, b, no_graph=kw.get('no_graph', False), no_control=kw.get('no_control', False))


def check_traj(y, ref, l1=L1_TOL, mx=1e-3):
    err = np.abs(y - ref)
    scale = max(1.0, np.abs(ref).max())
    assert err.mean() < l1 * scale, 'L1 %.3e' % err.mean()
    assert err.max() < mx * scale, 'max %.3e' % err.max()


@pytest.mark.parametrize('as_module', [True, False], ids=['device_resident', 'generic'])
@pytest.mark.parametrize('name', names('fixed_*.npz'))
def test_fixed_grid_golden(dev, name, as_module):
    from ndcn_amd import torchdiffeq as ode
    d = load_golden(name)
    f = make_func(d, dev, as_module)
    with torch.no_grad():
        y = ode.odeint(f, T(d['x0']).to(dev), T(d['t']).to(dev), method=name.split('_')[1])
    assert y.shape == d['traj'].shape
    assert np.array_equal(y[0].cpu().numpy(), d['x0'])
    check_traj(y.cpu().numpy(), d['traj'], l1=1e-5, mx=1e-4)


@pytest.mark.parametrize('as_module', [True, False], ids=['device_resident', 'generic'])
@pytest.mark.parametrize('name', names('dopri5_*.npz'))
def test_dopri5_golden(dev, name, as_module):
    from ndcn_amd import torchdiffeq as ode
    d = load_golden(name)
    f = make_func(d, dev, as_module, no_control='no_control' in name)
    log = []
    opts = {k[4:]: float(v) for k, v in d.items() if k.startswith('opt_')} or None
    with torch.no_grad():
        y = ode.odeint(f, T(d['x0']).to(dev), T(d['t']).to(dev), rtol=float(d['rtol']), atol=float(d['atol']),
                       method='dopri5', step_log=log, options=opts)
    nfe = dict([log.pop()])['nfe']
    ref = d['steplog']
    log = np.array(log)
    check_traj(y.cpu().numpy(), d['traj'], l1=1e-5, mx=2e-4)
    assert nfe == int(d['nfe']), (nfe, int(d['nfe']))
    assert np.array_equal(log[:, 2], ref[:, 2])                       # identical accept / reject sequence
    assert np.allclose(log[:, [0, 1, 4]], ref[:, [0, 1, 4]], rtol=1e-4)
    assert np.allclose(log[:, 3], ref[:, 3], rtol=5e-3, atol=1e-12)


@pytest.mark.parametrize('as_module', [True, False], ids=['module', 'callable'])
@pytest.mark.parametrize('name', names('adams_*.npz'))
def test_adams_golden(dev, name, as_module):
    """method='adams' (adams.py:62-170) on the HIP panel kernels - combine for predictor / corrector / phi differences,
    scale for the explicit phi, the error-ratio reduction - against the reference's own run: same orders, same accept /
    reject sequence, same number of evaluations, trajectory within the path's tolerance."""
    from ndcn_amd import torchdiffeq as ode
    d = load_golden(name)
    f = make_func(d, dev, as_module, no_control=bool(d['no_control']))
    log = []
    opts = {k[4:]: (int(v) if k == 'opt_max_order' else float(v)) for k, v in d.items() if k.startswith('opt_')} or None
    with torch.no_grad():
        y = ode.odeint(f, T(d['x0']).to(dev), T(d['t']).to(dev), rtol=float(d['rtol']), atol=float(d['atol']), method='adams',
                       options=opts, step_log=log)
    nfe = dict([log.pop()])['nfe']
    ref, got = d['steplog'], np.array(log)
    check_traj(y.cpu().numpy(), d['traj'], l1=1e-5, mx=2e-4)
    # every fixture, adams_tight (rtol 1e-5: 52 attempts, 105 evaluations) included: the error ratios are formed in ATen's
    # float32 order on reference-sized panels (rk.hip), so orders and accept / reject decisions are the reference's
    # (tools/micro/adams_tight_probe.py prints the first diverging attempt should that ever change)
    assert got.shape[0] == ref.shape[0] and nfe == int(d['nfe'])
    assert np.array_equal(got[:, 2:4], ref[:, 2:4])


@pytest.mark.parametrize('method', ['euler', 'midpoint', 'rk4', 'dopri5'])
def test_lattice_h256_solvers_against_oracle(dev, method):
    """H = 256 on a lattice: the device-resident solver runs rhs_fused3 (group-record plan) with the stage algebra of every
    method in its epilogues (RK4 stages, dopri5 COMBINE / ERROR) - trajectory and, for dopri5, the accept / reject log
    against the CPU oracle on the same inputs."""
    from ndcn_amd import torchdiffeq as ode, graphs, CsrOperator, _lib
    from ndcn_amd.neural_dynamics import ODEFunc
    side, H = 30, 256
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = CsrOperator.from_scipy(L, dev)
    torch.manual_seed(3)
    f = ODEFunc(H, A).to(dev).eval()
    x0 = torch.rand(side * side, H, generator=torch.Generator().manual_seed(4))
    t = torch.linspace(0., 1.2, 7)
    log = []
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), rtol=1e-3, atol=1e-4, method=method, step_log=log if method == 'dopri5' else None)
    assert A.rec is not None and _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
    Ao = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
    W, b = f.wt.weight.detach().cpu(), f.wt.bias.detach().cpu()
    ref_log = []
    ref = orc.odeint(lambda tt, xx: orc.odefunc_rhs(Ao, xx, W, b), x0, t, rtol=1e-3, atol=1e-4, method=method,
                     **({'step_log': ref_log} if method == 'dopri5' else {}))
    check_traj(y.cpu().numpy(), ref.numpy(), l1=1e-5, mx=2e-4)
    if method == 'dopri5':
        got = [r for r in log if r[0] != 'nfe']
        want = [r for r in ref_log if r[0] != 'nfe']
        assert len(got) == len(want) and [r[2] for r in got] == [r[2] for r in want]


@pytest.mark.parametrize('name', names('ndcn_*.npz'))
def test_ndcn_end_to_end_golden(dev, name):
    from ndcn_amd.neural_dynamics import NDCN
    d = load_golden(name)
    variant, method = name[len('ndcn_'):].rsplit('_', 1)
    A = orc.dense_from_csr(d['indptr'], d['indices'], d['data'], d['shape']).to(dev)    # the drivers' dense layout
    m = NDCN(input_size=1, hidden_size=1 if variant == 'no_embed' else 20, A=A, num_classes=1, dropout=0.0,
             no_embed=variant == 'no_embed', no_graph=variant == 'no_graph', no_control=variant == 'no_control',
             rtol=.01, atol=.001, method=method).to(dev)
    sd = {k[4:].replace('__', '.'): T(v) for k, v in d.items() if k.startswith('sd__')}
    m.load_state_dict(sd)                       # the reference's own key names
    with torch.no_grad():
        out = m(T(d['t']).to(dev), T(d['x0']).to(dev))
    assert out.shape == d['out'].shape
    check_traj(out.cpu().numpy(), d['out'], l1=1e-5, mx=2e-4)


@pytest.mark.parametrize('name', names('truth_*_coo.npz'))
def test_truth_dynamics_golden(dev, name):
    """Ground-truth generation of the three drivers (odeint defaults rtol 1e-7 / atol 1e-9, N x 1 state)."""
    from ndcn_amd import CsrOperator, hip
    from ndcn_amd import torchdiffeq as ode
    d = load_golden(name)
    n = int(d['n'])
    A = CsrOperator.from_arrays(d['A_indptr'], d['A_indices'], d['A_data'], (n, n), dev)
    L = CsrOperator.from_arrays(d['L_indptr'], d['L_indices'], d['L_data'], (n, n), dev)
    if 'heat' in name:
        f = lambda t, x: hip.spmm(L, x, alpha=-1.0)
    elif 'gene' in name:
        f = lambda t, x: hip.gene_rhs(A, x)
    else:
        f = lambda t, x: hip.mutual_rhs(A, x)
    log = []
    with torch.no_grad():
        y = ode.odeint(f, T(d['x0']).to(dev), T(d['t']).to(dev), method='dopri5', step_log=log)
    check_traj(y.cpu().numpy(), d['traj'], l1=2e-5, mx=2e-4)
    # the step sequence of the truth solve (rtol 1e-7 / atol 1e-9: every decision hangs on the last bits of the float32
    # norms and means, which the kernels form in ATen's order): as many attempts, the same accept / reject sequence and
    # the same number of evaluations as the reference-style solve of the oracle on the same inputs
    Ao = orc.coo_from_csr(d['A_indptr'], d['A_indices'], d['A_data'], (n, n))
    Lo = orc.coo_from_csr(d['L_indptr'], d['L_indices'], d['L_data'], (n, n))
    fo = {'heat': lambda t, x: orc.heat_rhs(Lo, x), 'gene': lambda t, x: orc.gene_rhs(Ao, x),
          'mutual': lambda t, x: orc.mutual_rhs(Ao, x)}[name.split('_')[1]]
    lo = []
    yo = orc.odeint(fo, T(d['x0']), T(d['t']), method='dopri5', step_log=lo)
    check_traj(yo.numpy(), d['traj'], l1=2e-5, mx=2e-4)              # (x ** h runs a different vector pow on another host CPU)
    nfe = dict([log.pop()])['nfe']
    assert len(log) == len(lo) and nfe == 2 + 6 * len(lo)
    assert [r[2] for r in log] == [r[2] for r in lo]
    if 'heat' in name:      # K1: closed form
        Ld = orc.dense_from_csr(d['L_indptr'], d['L_indices'], d['L_data'], (n, n)).numpy()
        exact = orc.heat_closed_form(Ld, d['x0'], d['t'])
        assert np.abs(y.cpu().numpy() - exact).mean() < 2e-5
        assert abs(float(y[-1].sum()) - float(d['x0'].sum())) < 5e-2      # K2 conservation


@pytest.mark.parametrize('name,H', [('cora', 64), ('pubmed', 16)])
def test_dgnn_block_golden(dev, name, H):
    """dgnn.py's differential_gcn hot path: ODEBlock2(ODEFunc(no_control), terminal=True) on the Planetoid topology."""
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc, ODEBlock2
    d = load_golden('dgnn_%s_H%d' % (name, H))
    g = load_golden('operators_' + name)
    n = int(g['n'])
    A = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    blk = ODEBlock2(ODEFunc(H, A, dropout=0.0, no_control=True), T(d['t']).to(dev), rtol=.1, atol=.1,
                    method='dopri5', terminal=True).to(dev).eval()
    with torch.no_grad():
        y = blk(T(d['x']).to(dev))
    assert y.shape == d['out'].shape
    check_traj(y.cpu().numpy(), d['out'], l1=1e-5, mx=2e-4)


def test_dgnn_block_pubmed_width_256_vs_oracle(dev):
    """Config C5 at the README / bench width: ODEBlock2(ODEFunc(no_control), 16 ticks on [0, 1.2], dopri5 rtol = atol = .1)
    on the Pubmed topology with H = 256 (dgnn.py:173-182; README.md:64 `--hidden 256`), every tick, against the oracle
    on the same seeded features - the solve bench.py --config C5 times on both sides - with the same accept / reject
    sequence (the reference fixture dgnn_pubmed_H16 covers the narrow width)."""
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc, ODEBlock2
    g = load_golden('operators_pubmed')
    n, H = int(g['n']), 256
    A = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    x = torch.rand(n, H, generator=torch.Generator().manual_seed(0))
    t = torch.linspace(0., 1.2, 16)
    func = ODEFunc(H, A, dropout=0.0, no_control=True)
    blk = ODEBlock2(func, t.to(dev), rtol=.1, atol=.1, method='dopri5', terminal=False).to(dev).eval()
    log = []
    with torch.no_grad():
        y = blk(x.to(dev))
        y2 = ode.odeint(func, x.to(dev), t.to(dev), rtol=.1, atol=.1, method='dopri5', step_log=log)
    assert torch.equal(y, y2)
    Ao = orc.coo_from_csr(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n))
    lo = []
    ref = orc.odeint(orc.OracleODEFunc(Ao, None, None, no_control=True), x, t, rtol=.1, atol=.1, method='dopri5', step_log=lo)
    nfe = dict([log.pop()])['nfe']
    assert [r[2] for r in log] == [r[2] for r in lo] and nfe == 2 + 6 * len(lo)
    check_traj(y.cpu().numpy(), ref.numpy(), l1=1e-5, mx=2e-4)


def test_rownorm_resblock_gcn_resgcn_golden(dev):
    """RowNorm, ResBlock (all four configurations), models.GCN and dgnn's resGCN Sequential on the
    Cora topology - reference outputs (fixtures G10) vs the drop-in modules on the HIP kernels, reference state_dicts
    loaded by key."""
    import torch.nn as nn
    from ndcn_amd import CsrOperator
    from ndcn_amd.ode_gcn import RowNorm, ResBlock
    from ndcn_amd.models import GCN
    d0 = load_golden('resgcn_rownorm')
    x = T(d0['x']).to(dev)
    with torch.no_grad():
        assert (RowNorm()(x).cpu() - T(d0['out'])).abs().max() <= 1e-6
        for tag in ('plain', 'norm', 'tv', 'euler', 'norm_tv'):
            d = load_golden('resgcn_block_' + tag)
            A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)
            blk = ResBlock(32, A, normalize='norm' in tag, time_varying='tv' in tag, Euler=tag == 'euler').to(dev).eval()
            sd = {}
            if 'W' in d:
                sd.update({'linear.weight': T(d['W']), 'linear.bias': T(d['b'])})
            if 'time_step' in d:
                sd['time_step'] = T(d['time_step'])
            blk.load_state_dict(sd)
            assert (blk(x).cpu() - T(d['out'])).abs().max() <= 2e-5, tag
        d = load_golden('resgcn_gcn')
        A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)
        gcn = GCN(32, 16, 7, dropout=0.5, num_middle_layers=1).to(dev).eval()
        gcn.load_state_dict({k[3:]: T(v) for k, v in d.items() if k.startswith('sd_')})
        assert (gcn(T(d['x']).to(dev), A).cpu() - T(d['out'])).abs().max() <= 2e-5
        for tag, norm, euler in (('model', False, False), ('model_norm_euler', True, True)):
            d = load_golden('resgcn_' + tag)
            A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)
            from ndcn_amd.neural_dynamics import _HipLinear
            model = nn.Sequential(_HipLinear(32, 32), nn.ReLU(inplace=True),
                                  *[ResBlock(32, A, dropout=0.5, normalize=norm, Euler=euler) for _ in range(2)],
                                  _HipLinear(32, 7)).to(dev).eval()
            model.load_state_dict({k[3:]: T(v) for k, v in d.items() if k.startswith('sd_')})
            assert (model(T(d['x']).to(dev)).cpu() - T(d['out'])).abs().max() <= 2e-5, tag


def test_dense_graph_convolution_golden(dev):
    """The dense-A GraphConvolution that dgnn.py's star import exposes (neural_dynamics.py:163-176): A (x W^T + b)
    flattened to 1 x (N * out), with and without bias - fixture from the reference (tools/gen_golden.py gen_gconv)."""
    from ndcn_amd.neural_dynamics import GraphConvolution
    d = load_golden('gconv_dense')
    A, x = T(d['A']).to(dev), T(d['x']).to(dev)
    for bias, wk, ok in ((True, 'W', 'out'), (False, 'W_nb', 'out_nb')):
        gc = GraphConvolution(7, 5, bias=bias).to(dev)
        sd = {'fc.weight': T(d[wk])}
        if bias:
            sd['fc.bias'] = T(d['b'])
        gc.load_state_dict(sd)                                   # the reference's state_dict keys
        with torch.no_grad():
            out = gc(x, A)
        assert tuple(out.shape) == tuple(d[ok].shape) == (1, A.shape[0] * 5)
        assert np.abs(out.cpu().numpy() - d[ok]).max() <= 1e-5
        out = gc(x, A)                                           # training path (autograd Functions): same values + grads
        assert np.abs(out.detach().cpu().numpy() - d[ok]).max() <= 1e-5
        out.sum().backward()
        assert gc.fc.weight.grad is not None and torch.isfinite(gc.fc.weight.grad).all()


@pytest.mark.parametrize('method', ['euler', 'midpoint', 'rk4'])
def test_hipgraph_replay_equals_eager(dev, method):
    """Fixed-grid steps replayed from one captured hipGraph (dt read from device memory) give bit-identical
    trajectories to eager launches, on equal and irregular time grids."""
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    d = load_golden('fixed_%s_irregular' % method)
    f = make_func(d, dev)
    x0, t = T(d['x0']).to(dev), T(d['t'])
    outs = []
    for use_graph in (False, True):
        s = DeviceSolver(f, x0.shape[0], method, use_graph=use_graph)
        s.begin(x0, float(t[0]))
        traj = [x0.clone()]
        for ti in t[1:].tolist():
            o = torch.empty_like(x0)
            s.advance(ti, o)
            traj.append(o)
        torch.cuda.synchronize()
        assert s.stats()['nfe'] == {'euler': 1, 'midpoint': 2, 'rk4': 4}[method] * (len(t) - 1)
        s.close()
        outs.append(torch.stack(traj))
    assert torch.equal(outs[0], outs[1])
    check_traj(outs[1].cpu().numpy(), d['traj'], l1=1e-5, mx=1e-4)


@pytest.mark.parametrize('case', ['H20', 'no_control_rec', 'no_control_pubmed'])
def test_dopri5_hipgraph_replay_equals_eager(dev, case):
    """One attempted dopri5 step replayed from a captured hipGraph (step size and dt * beta in device memory): step log
    and trajectory identical, bit for bit, to eager launches - generic kernels (H = 20), the group-record epilogue
    kernels (no_control, H = 256, lattice) and the Pubmed topology (no plan: row SpMM + stage kernels)."""
    from ndcn_amd import CsrOperator, graphs
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    if case == 'H20':
        d = load_golden('dopri5_loose')
        f = make_func(d, dev)
        x0, t, rtol, atol = T(d['x0']).to(dev), T(d['t']), float(d['rtol']), float(d['atol'])
    else:
        if case == 'no_control_rec':
            A = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(60)), dev)
            n = 3600
        else:
            g = load_golden('operators_pubmed')
            n = int(g['n'])
            A = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
        torch.manual_seed(1)
        f = ODEFunc(256, A, no_control=True).to(dev).eval()
        x0 = torch.rand(n, 256, device=dev)
        t, rtol, atol = torch.linspace(0., 1.2, 16), .1, .1
        if case == 'no_control_rec':
            A.ensure_plans(256)
            assert A.rec is not None
            t, rtol, atol = torch.linspace(0., 3., 9), 1e-3, 1e-4       # a few dozen attempts
    outs, logs = [], []
    for use_graph in (False, True):
        s = DeviceSolver(f, x0.shape[0], 'dopri5', rtol, atol, use_graph=use_graph)
        s.begin(x0, float(t[0]))
        traj = [x0.clone()]
        for ti in t[1:].tolist():
            o = torch.empty_like(x0)
            s.advance(ti, o)
            traj.append(o)
        torch.cuda.synchronize()
        logs.append((s.steplog(), s.stats()['nfe']))
        s.close()
        outs.append(torch.stack(traj))
    assert logs[0] == logs[1] and len(logs[0][0]) >= 2
    assert torch.equal(outs[0], outs[1])
    # the whole time vector in ONE call (ticks of a step evaluated together): bit-identical to tick-by-tick
    s = DeviceSolver(f, x0.shape[0], 'dopri5', rtol, atol)
    s.begin(x0, float(t[0]))
    many = torch.empty((len(t) - 1,) + tuple(x0.shape), device=dev)
    s.advance_many(t[1:].tolist(), many)
    torch.cuda.synchronize()
    assert (s.steplog(), s.stats()['nfe']) == logs[0]
    s.close()
    assert torch.equal(many, outs[0][1:])
    if case == 'H20':
        check_traj(outs[1].cpu().numpy(), d['traj'], l1=1e-5, mx=2e-4)


def test_borrowed_initial_state_equals_the_copied_one(dev):
    """ndcn_solver_begin_borrowed (ABI 9): dopri5 reads y0 where the caller keeps it - odeint() hands over the first
    panel of its solution - until the panel would come up for writing; same launches, same results, y0 untouched; an
    output that overlaps the borrowed panel is refused; restarts (fewer than two accepted steps, then many) stay exact."""
    from ndcn_amd import _lib, graphs
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    A = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(40)), dev)
    torch.manual_seed(4)
    f = ODEFunc(256, A).to(dev).eval()
    x0 = torch.rand(1600, 256, device=dev)
    keep = x0.clone()
    ticks = [0.05, 0.4, 1.5, 4.0, 9.0]                       # the first two fall into the first accepted step
    res = {}
    for borrow in (False, True):
        s = DeviceSolver(f, 1600, 'dopri5', .01, .001)
        outs = []
        for rep in range(3):                                 # restarts: after one tick (no rotation yet), then full solves
            s.begin(x0, 0.0, borrow=borrow)
            o = torch.empty((len(ticks),) + tuple(x0.shape), device=dev)
            if rep == 0:
                s.advance(ticks[0], o[0])
            else:
                s.advance_many(ticks, o)
            torch.cuda.synchronize()
            outs.append((o[:1].clone() if rep == 0 else o, s.steplog()))
        if borrow:
            s.begin(x0, 0.0, borrow=True)
            with pytest.raises(_lib.NdcnHipError, match='overlaps the initial state'):
                s.advance(1.0, x0)
            big = torch.empty((3,) + tuple(x0.shape), device=dev)
            big[1].copy_(x0)
            s.begin(big[1], 0.0, borrow=True)
            with pytest.raises(_lib.NdcnHipError, match='overlaps the initial state'):
                s.advance_many([1.0, 2.0], big[:2])
            s.advance_many([1.0], big[2:])                   # the panel right behind it is fine
        s.close()
        res[borrow] = outs
    assert torch.equal(x0, keep)
    assert len(res[True][1][1]) >= 4                         # several accepted steps: the caller's panel left the rotation
    for (a, la), (b, lb) in zip(res[False], res[True]):
        assert la == lb and torch.equal(a, b)
    assert torch.equal(res[True][1][0], res[True][2][0])


@pytest.mark.parametrize('side', [48, 55, 64])       # 48: some workgroups of the persistent grid get no tile
def test_fused_epilogue_solver_equals_generic_path(dev, side):
    """H = 256: the device-resident solver runs the stage algebra / error norm inside the fused RHS epilogue
    (rhs_fused2.hip); the generic path runs the same RHS kernel plus the separate rk.hip kernels.  Same
    summation orders on both sides, so trajectories and step logs must agree to rounding of the controller."""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    A = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(side)), dev)     # 55^2 = 3025: ragged tail
    torch.manual_seed(2)
    f = ODEFunc(256, A).to(dev).eval()
    x0 = torch.rand(side * side, 256, device=dev)
    with torch.no_grad():
        for method, t in (('euler', torch.linspace(0., 1., 5)), ('rk4', torch.linspace(0., 2., 5)),
                          ('dopri5', torch.tensor([0., 0.4, 1.5, 3.0]))):
            la, lb = [], []
            ya = ode.odeint(f, x0, t.to(dev), rtol=.01, atol=.001, method=method, step_log=la)
            yb = ode.odeint(lambda tt, y: f(tt, y), x0, t.to(dev), rtol=.01, atol=.001, method=method, step_log=lb)
            assert float((ya - yb).abs().max()) <= 1e-5 * float(yb.abs().max())
            if method == 'rk4':
                assert torch.equal(ya, yb)        # stage algebra in the epilogue == the separate stage kernels, bit for bit
            if method == 'dopri5':
                assert la[-1] == lb[-1]                                   # same number of RHS evaluations
                assert [r[2] for r in la[:-1]] == [r[2] for r in lb[:-1]]
                assert np.allclose([r[3] for r in la[:-1]], [r[3] for r in lb[:-1]], rtol=1e-4)
        # oracle check on the smaller case
        if side == 55:
            L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
            Ao = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
            fo = orc.OracleODEFunc(Ao, f.wt.weight.detach().cpu(), f.wt.bias.detach().cpu())
            ref = orc.odeint(fo, x0.cpu(), torch.tensor([0., 0.4, 1.5, 3.0]), rtol=.01, atol=.001, method='dopri5')
            check_traj(ya.cpu().numpy(), ref.numpy(), l1=1e-5, mx=2e-4)


def test_sharded_path_single_rank_equals_device_solver(dev):
    """The multi-GPU code path (ndcn_amd/sharding.py: halo plan, RCCL all-to-all-v, global reductions, Python
    stepping over ndcn_rhs_rk_f32) run with ONE rank over the nccl backend must reproduce the device-resident
    solver: same kernels, same order.  (Two-rank correctness is covered on CPU by tests/test_sharding_gloo.py.)"""
    import torch.distributed as dist
    from ndcn_amd import graphs, sharding, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1, device_id=dev)
    try:
        side, H = 48, 256
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
        torch.manual_seed(4)
        f = ODEFunc(H, graphs.to_device(L, dev)).to(dev).eval()
        x0 = torch.rand(side * side, H, device=dev)
        t = torch.tensor([0., 0.7, 2.0], device=dev)
        plan = sharding.HaloPlan(L, [0, side * side], 0, dev)
        assert plan.n_halo == 0
        with torch.no_grad():
            la, lb = [], []
            ya = ode.odeint(f, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=la)
            yb = sharding.sharded_odeint(hip, f, plan, side * side, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lb)
        assert [r[2] for r in la[:-1]] == [r[2] for r in lb if r[0] != 'nfe']
        assert float((ya - yb).abs().max()) <= 1e-5 * float(ya.abs().max())
        # RCCL's all-to-all-v actually executes: the self-halo hook routes the first 2 lattice rows through the exchange
        # (non-empty self split), on the side stream, overlapped with the interior launch
        plan2 = sharding.HaloPlan(L, [0, side * side], 0, dev, self_halo=2 * side)
        assert plan2.n_halo == 2 * side and plan2.send_counts == [2 * side] and plan2.ranges is not None
        with torch.no_grad():
            lc = []
            yc = sharding.sharded_odeint(hip, f, plan2, side * side, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lc)
            yr = sharding.sharded_odeint(hip, f, plan2, side * side, x0, torch.linspace(0., 1., 4).to(dev), method='rk4')
            y4 = ode.odeint(f, x0, torch.linspace(0., 1., 4).to(dev), method='rk4')
        assert [r[2] for r in lc if r[0] != 'nfe'] == [r[2] for r in lb if r[0] != 'nfe']
        assert float((yc - yb).abs().max()) <= 1e-5 * float(ya.abs().max())
        assert float((yr - y4).abs().max()) <= 1e-5 * float(y4.abs().max())
        # ---- the same shards through the C ABI: RCCL communicator + halo plan + sharded DEVICE-RESIDENT solver
        # (ndcn_comm_* / ndcn_halo_plan_* / ndcn_solver_desc::shard): no Python between the evaluations.  Row-split form
        # (lattice band), two-phase form (scattered self-halo), one-launch form (no halo): each must reproduce the
        # Python-stepped sharded solve - same kernels, same order - and the unsharded solver within the path's tolerance.
        from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
        plan3 = sharding.HaloPlan(L, [0, side * side], 0, dev, self_halo='scatter:500')
        assert plan3.two_phase is not None and plan3.n_halo == 500
        tt = [0.7, 2.0]
        for pl in (plan2, plan3, plan):
            shard = sharding.DeviceShard(pl, side * side)
            solver = DeviceSolver(f, side * side, 'dopri5', .01, .001, shard=shard)
            out = torch.empty(2, side * side, H, device=dev)
            with torch.no_grad():
                solver.begin(x0, 0.0)
                solver.advance_many(tt, out)
                lp = []
                yp = sharding.sharded_odeint(hip, f, pl, side * side, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lp)
            torch.cuda.synchronize()
            ld = solver.steplog()
            assert [r[2] for r in ld] == [r[2] for r in lp if r[0] != 'nfe'] == [r[2] for r in la[:-1]]
            assert float((out - yp[1:]).abs().max()) <= 1e-6 * float(ya.abs().max())
            assert float((out - ya[1:]).abs().max()) <= 1e-5 * float(ya.abs().max())
            assert int(solver.stats()['nfe']) == dict([lp[-1]])['nfe']
            # fixed grid on the shard: rk4 with the stage algebra in the split launches
            rk = DeviceSolver(f, side * side, 'rk4', shard=shard)
            o4 = torch.empty(3, side * side, H, device=dev)
            with torch.no_grad():
                rk.begin(x0, 0.0)
                rk.advance_many(torch.linspace(0., 1., 4)[1:].tolist(), o4)
            torch.cuda.synchronize()
            assert float((o4 - y4[1:]).abs().max()) <= 1e-5 * float(y4.abs().max())
            solver.close(); rk.close(); shard.close()
        # bench runner of the sharded path (with the hook: exchange timing is recorded)
        runner = sharding.ShardedGridBench(f, 48, 1, 0, dev, 5.0, .01, .001)
        assert runner.run_steps(3) == 3 and runner.nfe() >= 2 + 18
        os.environ['NDCN_SELF_HALO'] = str(2 * side)
        try:
            runner = sharding.ShardedGridBench(f, 48, 1, 0, dev, 5.0, .01, .001)
            runner.func.timing = {}
            assert runner.run_steps(3) == 3
            torch.cuda.synchronize()
            tm = runner.func.drain_timing()
            assert tm['n'] >= 10 and tm['exchange_us'] > 0 and tm['interior_us'] > 0
        finally:
            del os.environ['NDCN_SELF_HALO']
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('H', [256, 20])
def test_sharded_training_single_rank_self_halo_equals_unsharded_training(dev, H):
    """Training through the sharded path on the device (round 6): the halo exchange and its backward - the reverse all-to-all-v over
    RCCL, accumulated by the scatter operator - with ONE rank whose self-halo hook routes own rows through the exchange; the local
    right-hand side on [own | halo] through autograd_ops.  Against training through the unsharded solver (itself pinned to the oracle's
    gradients in test_gpu_autograd.py): rk4 - the same gradient; dopri5 - the sharded form keeps the controller's step sizes as
    constants, so it is compared at a tolerance where that choice does not matter (and the forward solve is the same solve)."""
    import torch.distributed as dist
    from ndcn_amd import graphs, sharding, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29643', rank=0, world_size=1, device_id=dev)
    try:
        side = 24
        n = side * side
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
        torch.manual_seed(4)
        f = ODEFunc(H, graphs.to_device(L, dev)).to(dev)
        x_init = torch.rand(n, H, generator=torch.Generator().manual_seed(5)).to(dev)
        t = torch.linspace(0., 1., 5).to(dev)
        wgt = torch.randn(5, n, H, generator=torch.Generator().manual_seed(6)).to(dev)
        for spec in (2 * side, 'scatter:150'):
            plan = sharding.HaloPlan(L, [0, n], 0, dev, self_halo=spec)
            assert plan.n_halo > 0 and sum(plan.send_counts) == plan.n_halo
            sub = 16
            fine = torch.cat([torch.linspace(float(t[i]), float(t[i + 1]), sub + 1)[:-1] for i in range(len(t) - 1)] + [t[-1:].cpu()]).to(dev)
            for method, kw, tol in (('rk4', {}, 2e-5), ('dopri5', dict(rtol=1e-6, atol=1e-8), 3e-2)):      # (dopri5 measured: 3e-3 .. 1e-2 of the gradient's scale, depending on the step sequence)
                res = []
                for sharded in (True, False):
                    for p_ in f.parameters():
                        p_.grad = None
                    x0 = x_init.clone().requires_grad_(True)
                    if sharded:
                        st = {}
                        y = sharding.sharded_odeint(hip, f, plan, n, x0, t, method=method, stats=st, **kw)
                        assert st['form'] == 'one_launch+autograd'
                    elif method == 'dopri5':
                        # the yardstick for the frozen-controller gradient: the exact flow's (RK4 on a 16 x finer grid, unsharded) - the
                        # unsharded dopri5 tape follows the reference THROUGH the controller and sits several per cent from both
                        y = ode.odeint(f, x0, fine, method='rk4')[::sub]
                    else:
                        y = ode.odeint(f, x0, t, method=method, **kw)
                    (y * wgt).sum().backward()
                    if sharded:
                        sharding.allreduce_gradients(f.parameters())
                    res.append((y.detach(), x0.grad.clone(), f.wt.weight.grad.clone(), f.wt.bias.grad.clone()))
                (ya, *ga), (yb, *gb) = res
                assert float((ya - yb).abs().max()) <= (2e-5 if method == 'rk4' else 1e-4) * float(yb.abs().max())
                for a, b in zip(ga, gb):
                    r = float((a - b).abs().max() / b.abs().max())
                    assert r < tol, (spec, method, r)
    finally:
        dist.destroy_process_group()


def _two_rank_graph(case):
    from ndcn_amd import graphs
    if case == 'grid':
        R, C = 60, 40                                          # 2400 nodes, 30 lattice rows per rank
        return graphs.normalized_laplacian(graphs.grid_8_neighbor_rect(R, C)), [0, 30 * C, R * C]
    if case == 'power_law':                                    # hubs: the long-row plan next to the halo panel
        full = graphs.normalized_laplacian(graphs.make_graph('power_law', 4000, seed=2))
        return full, [0, 2000, 4000]
    full = graphs.normalized_laplacian(graphs.make_graph('small_world', 3000, seed=3))    # random shortcuts: wide halo
    return full, [0, 1500, 3000]


def _two_rank_worker(rank, world, port, case, ret):
    import sys
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ndcn_amd import graphs, sharding, hip
        from ndcn_amd.neural_dynamics import ODEFunc
        dev = torch.device('cuda:0')
        H = 256
        torch.manual_seed(0)
        f = ODEFunc(H, None).to(dev)
        full, bounds = _two_rank_graph(case)
        n = full.shape[0]
        plan = sharding.HaloPlan(full[bounds[rank]:bounds[rank + 1]], bounds, rank, dev)
        x = torch.rand(n, H, generator=torch.Generator().manual_seed(1))
        xl = x[bounds[rank]:bounds[rank + 1]].contiguous().to(dev)
        t = torch.linspace(0., 1.5, 4).to(dev)
        out = {'halo': plan.n_halo}
        if case == 'power_law':
            plan.local_op.ensure_plans(H)
            hub = plan.local_op.hub
            out['hubs'] = 0 if hub is None else hub['n']
            # (Barabasi-Albert: the early nodes are the hubs - rank 0 owns them; rank 1's rows may have none)
            assert hub is None or hub['halo_S'].shape[0] == plan.n_halo + hub['n']
        with torch.no_grad():
            for method in ('rk4', 'dopri5'):
                log = []
                y = sharding.sharded_odeint(hip, f, plan, n, xl, t, rtol=1e-3, atol=1e-4, method=method, step_log=log)
                out[method] = y.cpu().numpy()
                out[method + '_log'] = [r for r in log if r[0] != 'nfe']
        from ndcn_amd import _lib
        out['path'] = int(_lib.load().ndcn_debug_last_rhs_path())
        out['W'], out['b'] = f.wt.weight.detach().cpu().numpy(), f.wt.bias.detach().cpu().numpy()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['grid', 'small_world', 'power_law'])
def test_two_rank_sharded_hip_path_equals_device_solver(dev, case):
    """The N > 1 product path with world size 2 ON THE GPU: two processes (both on cuda:0, gloo with host staging
    because RCCL refuses two ranks on one device) run HaloPlan + halo exchange + the HALO variants of the fused
    RHS+RK kernel + the global error reduction; the stitched trajectory must equal the single-process
    device-resident solver on the whole graph, with the same accept / reject sequence on both ranks."""
    import torch.multiprocessing as mp
    from ndcn_amd import graphs, CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq import odeint
    world, port = 2, 29900 + os.getpid() % 90 + {'grid': 0, 'small_world': 1, 'power_law': 2}[case]
    # (a SPAWNED manager: a fork()ed server inherits this process's garbage - tensors, events, handles of a HIP context it does
    # not have - and its garbage collector then runs their destructors)
    ret = mp.get_context('spawn').Manager().dict()
    mp.spawn(_two_rank_worker, args=(world, port, case, ret), nprocs=world, join=True)
    assert len(ret) == world and ret[0]['halo'] > 0 and ret[1]['halo'] > 0
    if case == 'power_law':                                    # the long-row plan was active on the shards, beside their halo panels
        from ndcn_amd import _lib
        assert ret[0]['hubs'] > 0
        assert ret[0]['path'] == _lib.PATH_FUSED2 | _lib.PATH_HUB | _lib.PATH_HALO, ret[0]['path']
    H = 256
    full, _ = _two_rank_graph(case)
    f = ODEFunc(H, CsrOperator.from_scipy(full, dev)).to(dev)
    f.load_state_dict({'wt.weight': torch.from_numpy(ret[0]['W']), 'wt.bias': torch.from_numpy(ret[0]['b'])})
    x = torch.rand(full.shape[0], H, generator=torch.Generator().manual_seed(1)).to(dev)
    t = torch.linspace(0., 1.5, 4).to(dev)
    assert ret[0]['dopri5_log'] == ret[1]['dopri5_log'] and len(ret[0]['dopri5_log']) >= 3
    with torch.no_grad():
        for method in ('rk4', 'dopri5'):
            ref = odeint(f, x, t, rtol=1e-3, atol=1e-4, method=method).cpu().numpy()
            got = np.concatenate([ret[r][method] for r in range(world)], axis=1)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() < 2e-5, method


@pytest.mark.parametrize('config', ['M', 'C4'])
def test_bench_two_ranks_on_one_device(dev, config):
    """bench.py's N > 1 flow end to end (torchrun, sharded runner, barriers, max over ranks, one JSON line from rank 0)
    with two ranks on the one device of the test box (gloo hook); the driver runs it with nccl on 2/4/8 GPUs.  M: the
    metric's grid (each rank builds its own lattice rows), started as plain `python bench.py --gpus 2` - bench.py launches
    its own ranks; C4: the small world of BASELINE config 4 (scattered halo: two-phase evaluation) under an explicit
    torch.distributed.run, the form the driver uses."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, NDCN_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1', NDCN_C4_NODES='6000')
    import socket
    with socket.socket() as sk:                                       # a port nobody holds (a fixed one collided once in a full run)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    launcher = [] if config == 'M' else ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                                         '127.0.0.1', '--master-port', str(port)]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--side', '96', '--steps', '6',
                                         '--warmup', '2', '--config', config]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 6 and out['value'] > 0 and out['scaling'] == 'weak'
    assert out['halo_exchange']['bytes_received_per_rhs_per_gpu'] > 0 and out['cpu_baseline'] is None
    assert out['halo_exchange']['two_phase_own_columns_under_exchange'] == (config == 'C4')
    assert out['roofline']['kernel'] == 'rhs_fused' and out['roofline']['traffic'] is None


def test_tuple_state_generic_path(dev):
    from ndcn_amd import torchdiffeq as ode

    def f(t, y):
        a, b = y
        return (-a * t + b.mean(), torch.sin(t) * b - a.sum() * 0.01)
    g = torch.Generator().manual_seed(3)
    y0 = (torch.rand(7, 3, generator=g), torch.rand(5, generator=g))
    t = torch.linspace(0., 2., 9)
    for method in ('euler', 'rk4', 'dopri5'):
        ref = orc.odeint(f, y0, t, rtol=1e-4, atol=1e-6, method=method)
        got = ode.odeint(f, tuple(v.to(dev) for v in y0), t.to(dev), rtol=1e-4, atol=1e-6, method=method)
        for gx, r in zip(got, ref):
            assert (gx.cpu() - r).abs().max() < 1e-4


def test_error_behaviour_on_device(dev):
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd import graphs
    A = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(8)), dev)
    f = ODEFunc(8, A).to(dev).eval()
    x = torch.rand(64, 8).to(dev)
    with torch.no_grad():
        with pytest.raises(AssertionError):
            ode.odeint(f, x, torch.tensor([0., 1., 0.5]).to(dev), method='dopri5')
        bad = x.clone(); bad[3, 3] = float('nan')
        with pytest.raises(AssertionError):
            ode.odeint(f, bad, torch.tensor([0., 1.]).to(dev), method='dopri5')
        with pytest.raises(AssertionError):
            ode.odeint(lambda t, y: y * float('inf'), x, torch.tensor([0., 1.]).to(dev), method='dopri5')


def test_large_grid_properties(dev):
    """Full-size properties at the metric's case (1M-node grid, H = 256), where the oracle cannot run a whole
    solve in seconds: (i) SpMM linearity, (ii) row-sum identity of the normalised Laplacian on D^1/2 1,
    (iii) device-resident and generic dopri5 agree, (iv) a sampled row block equals the fp64 oracle."""
    from ndcn_amd import graphs, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    S, H = 1000, 256
    Agrid = graphs.grid_8_neighbor(S)
    L = graphs.normalized_laplacian(Agrid)
    A = graphs.to_device(L, dev)
    torch.manual_seed(0)
    X = torch.rand(S * S, H, device=dev)
    Z = torch.rand(S * S, H, device=dev)
    lin = hip.spmm(A, X + 2 * Z) - (hip.spmm(A, X) + 2 * hip.spmm(A, Z))
    assert float(lin.abs().max()) < 1e-4
    # L (D^1/2 1) = 0 for the normalised Laplacian
    deg = torch.from_numpy(np.asarray(Agrid.sum(1)).reshape(-1).astype(np.float32))
    v = deg.sqrt().to(dev).view(-1, 1).repeat(1, 4).contiguous()
    assert float(hip.spmm(A, v).abs().max()) < 1e-5
    # sampled rows vs fp64
    rows = np.r_[0:64, 500000:500064, S * S - 64:S * S]
    sub = L[rows]
    got = hip.spmm(A, X)[torch.from_numpy(rows).to(dev)].cpu().numpy()
    ref = orc.spmm_f64(sub.indptr, sub.indices, sub.data, X.cpu().numpy())
    assert np.abs(got - ref).max() < 1e-5
    del Z, lin
    # the FUSED right-hand side at the metric's size (rhs_fused3: the launch bench.py times) against fp64 on sampled rows - first, middle
    # and last patches of the lattice plan - directly, not through another HIP path; then the same rows out of a launch that carries
    # an RK epilogue (the dopri5 stage-3 form): K the same bits, y_next = y0 + c1 k1 + c2 K against fp64
    f = ODEFunc(H, A).to(dev).eval()
    rows = np.r_[0:64, 1000:1032, 499968:500064, S * S - 64:S * S]
    _sampled_rhs_check(L, A, f, X, dev, rows)
    from ndcn_amd import _lib
    assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
    ridx = torch.from_numpy(rows).to(dev)
    k1 = torch.rand(S * S, H, device=dev)
    y0 = torch.rand(S * S, H, device=dev)
    K0 = hip.rhs(A, X, f.wt.weight, f.wt.bias)[ridx]
    K, yn = hip.rhs_rk(A, X, f.wt.weight.detach(), f.wt.bias.detach(), 'combine', y0, [k1], [0.075, 0.225])
    assert torch.equal(K[ridx], K0)
    want = y0[ridx].double() + (0.075 * k1[ridx].double() + 0.225 * K0.double())
    assert float((yn[ridx].double() - want).abs().max()) < 4e-7 * float(want.abs().max())
    del k1, y0, K, yn
    # solver agreement at full size
    x0 = torch.rand(S * S, H, device=dev)
    t = torch.tensor([0., 0.6], device=dev)
    with torch.no_grad():
        la, lb = [], []
        ya = ode.odeint(f, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=la)
        yb = ode.odeint(lambda tt, y: f(tt, y), x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lb)
    assert la[-1] == lb[-1]
    assert float((ya[-1] - yb[-1]).abs().max()) < 1e-4


def _sampled_rhs_check(L, A, f, X, dev, rows):
    """relu(W (A X) + b) on a sample of rows against fp64."""
    from ndcn_amd import hip
    sub = L[rows]
    got = hip.rhs(A, X, f.wt.weight, f.wt.bias)[torch.from_numpy(rows).to(dev)].cpu().double().numpy()
    Xh = X.cpu().numpy()
    S = orc.spmm_f64(sub.indptr, sub.indices, sub.data, Xh)
    W = f.wt.weight.detach().cpu().double().numpy()
    ref = np.maximum(S @ W.T + f.wt.bias.detach().cpu().double().numpy(), 0)
    # bound: the documented error of the fp32-grade products (split16.h: 2e-7 of the sum of magnitudes; the fma chain of the
    # fold the same per entry) against the magnitudes that actually enter each output: sum_k |W_ok| sum_j |a_ij| |x_jk|
    absS = orc.spmm_f64(sub.indptr, sub.indices, np.abs(sub.data), np.abs(Xh))
    mag = absS @ np.abs(W).T + np.abs(f.wt.bias.detach().cpu().double().numpy())
    assert (np.abs(got - ref) <= 2e-6 * mag + 1e-30).all(), float((np.abs(got - ref) / (mag + 1e-30)).max())


@pytest.mark.parametrize('layout', [None, 'degree'])
def test_config_c2_full_size_properties(dev, layout):
    """BASELINE config 2 at full size (100k-node G(n,p), mean degree 39.9, H = 256, RK4 on linspace(0,5,100)), where the
    oracle cannot run the solve in seconds: sampled RHS rows against fp64, SpMM linearity, the device-resident solver
    (stage algebra in the RHS epilogues) bit-equal to the generic path (separate stage kernels) over the first steps, and
    the --layout relabelling leaves the solution unchanged up to the permutation."""
    from ndcn_amd import graphs, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    n, H = 100000, 256
    G0 = graphs.make_graph('random', n, seed=0)
    G = graphs.reorder_nodes(G0, layout)
    L = graphs.normalized_laplacian(G)
    assert abs(L.nnz / n - 40.9) < 0.5
    A = graphs.to_device(L, dev)
    torch.manual_seed(0)
    f = ODEFunc(H, A).to(dev).eval()
    X = torch.rand(n, H, device=dev)
    Z = torch.rand(n, H, device=dev)
    assert float((hip.spmm(A, X + 2 * Z) - (hip.spmm(A, X) + 2 * hip.spmm(A, Z))).abs().max()) < 2e-4
    _sampled_rhs_check(L, A, f, X, dev, np.r_[0:48, 50000:50048, n - 48:n])
    t = torch.linspace(0., 5., 100)[:4].to(dev)
    with torch.no_grad():
        ya = ode.odeint(f, X, t, method='rk4')
        yb = ode.odeint(lambda tt, y: f(tt, y), X, t, method='rk4')
    assert torch.equal(ya, yb)
    if layout is not None:                       # P A P^T on P x: the same trajectory, rows permuted
        new = torch.from_numpy(graphs.node_mapping(G0, layout)).to(dev)
        f0 = ODEFunc(H, graphs.to_device(graphs.normalized_laplacian(G0), dev)).to(dev).eval()
        f0.load_state_dict(f.state_dict())
        X0 = torch.empty_like(X)
        X0[:] = X[new]                           # node i of the original graph sits at position new[i]
        with torch.no_grad():
            y0 = ode.odeint(f0, X0, t, method='rk4')
        assert float((y0[-1] - ya[-1][new]).abs().max()) < 1e-4 * float(ya[-1].abs().max())


def test_config_c3_full_size_properties(dev):
    """BASELINE config 3 at full size (1M-node Barabasi-Albert m = 5, H = 256, dopri5): the long-row plan is active,
    sampled RHS rows (hubs included) agree with fp64, the device-resident solver agrees with the generic path (same
    accept / reject decisions), and the truth dynamics of the config (mutualistic, edge-wise kernel) run at this size."""
    from ndcn_amd import graphs, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    n, H = 1000000, 256
    G = graphs.make_graph('power_law', n, seed=0)
    L = graphs.normalized_laplacian(G)
    deg = np.diff(L.indptr)
    assert deg.max() > 1000 and abs(L.nnz - 11e6) < 1e5
    A = graphs.to_device(L, dev)
    torch.manual_seed(0)
    f = ODEFunc(H, A).to(dev).eval()
    X = torch.rand(n, H, device=dev)
    hubs = np.argsort(-deg)[:16]
    _sampled_rhs_check(L, A, f, X, dev, np.unique(np.r_[hubs, 0:32, 500000:500032, n - 32:n]))
    assert A.hub is not None and A.hub['n'] > 1000
    t = torch.tensor([0., 0.5], device=dev)
    with torch.no_grad():
        la, lb = [], []
        ya = ode.odeint(f, X, t, rtol=.01, atol=.001, method='dopri5', step_log=la)
        yb = ode.odeint(lambda tt, y: f(tt, y), X, t, rtol=.01, atol=.001, method='dopri5', step_log=lb)
    assert la[-1] == lb[-1] and [r[2] for r in la[:-1]] == [r[2] for r in lb[:-1]]
    assert float((ya[-1] - yb[-1]).abs().max()) < 1e-4 * max(1.0, float(yb[-1].abs().max()))
    del ya, yb, X
    # truth RHS of the config at full size: finite, and linear-response sanity of the edge-wise kernel on a constant state
    Aadj = graphs.to_device(G, dev)
    x = torch.full((n, 1), 2.0, device=dev)
    out = hip.mutual_rhs(Aadj, x)
    want = 0.1 + 2.0 * (1 - 2.0 / 5) * (2.0 / 1 - 1) + (deg - 1) * (2.0 * 2.0 / (5 + 0.9 * 2.0 + 0.1 * 2.0))
    assert np.abs(out.cpu().numpy().reshape(-1) - want.astype(np.float32)).max() < 1e-3 * want.max()


def test_config_c4_share_full_size_properties(dev):
    """One GPU's share of BASELINE config 4 (500k-node Newman-Watts-Strogatz small world k = 5 p = 0.5, H = 256, dopri5;
    the 8-GPU run shards a 4M-node graph of the same generator): sampled RHS rows against fp64, SpMM linearity, the
    device-resident solver agrees with the generic path (same accept / reject decisions), and the truth dynamics of the
    config (gene regulation, edge-wise kernel) run at this size."""
    from ndcn_amd import graphs, hip
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    n, H = 500000, 256
    G = graphs.make_graph('small_world', n, seed=0)
    L = graphs.normalized_laplacian(G)
    deg = np.diff(L.indptr)
    assert 5.5 < L.nnz / n - 1 < 6.1 and deg.max() < 64
    A = graphs.to_device(L, dev)
    torch.manual_seed(0)
    f = ODEFunc(H, A).to(dev).eval()
    X = torch.rand(n, H, device=dev)
    Z = torch.rand(n, H, device=dev)
    assert float((hip.spmm(A, X + 2 * Z) - (hip.spmm(A, X) + 2 * hip.spmm(A, Z))).abs().max()) < 1e-4
    del Z
    _sampled_rhs_check(L, A, f, X, dev, np.r_[0:48, 250000:250048, n - 48:n])
    t = torch.tensor([0., 0.5], device=dev)
    with torch.no_grad():
        la, lb = [], []
        ya = ode.odeint(f, X, t, rtol=.01, atol=.001, method='dopri5', step_log=la)
        yb = ode.odeint(lambda tt, y: f(tt, y), X, t, rtol=.01, atol=.001, method='dopri5', step_log=lb)
    assert la[-1] == lb[-1] and [r[2] for r in la[:-1]] == [r[2] for r in lb[:-1]]
    assert float((ya[-1] - yb[-1]).abs().max()) < 1e-4 * max(1.0, float(yb[-1].abs().max()))
    del ya, yb, X
    # truth RHS of the config on a constant state x = 2: -b x^f + sum_j A_ij x^h / (x^h + 1) with b = f = 1, h = 2
    x = torch.full((n, 1), 2.0, device=dev)
    out = hip.gene_rhs(graphs.to_device(G, dev), x)
    want = -2.0 + (deg - 1) * (4.0 / 5.0)
    assert np.abs(out.cpu().numpy().reshape(-1) - want.astype(np.float32)).max() < 1e-4 * np.abs(want).max()


@pytest.mark.parametrize('kind', ['heat', 'gene', 'mutualistic'])
def test_driver_counterpart_trains(dev, kind, capsys):
    """the build's driver counterpart (flags, split, Adam/L1 loop, log format) runs config C1 end to
    end on the HIP path and the training loss goes down."""
    from ndcn_amd.drivers.dynamics import main
    out = main(kind, ['--network', 'grid', '--sampled_time', 'equal', '--baseline', 'ndcn', '--gpu', '0',
                      '--niters', '30', '--test_freq', '10', '--time_tick', '20', '--method', 'euler'])
    text = capsys.readouterr().out
    lines = [l for l in text.splitlines() if l.startswith('Iter ')]
    assert len(lines) == 4 and 'Train Loss' in lines[0] and 'Test Loss' in lines[0] and '| Time ' in lines[0]
    first = float(lines[0].split('Train Loss ')[1].split('(')[0])
    last = float(lines[-1].split('Train Loss ')[1].split('(')[0])
    assert last < first
    assert out['params'] == 901                                     # 901 parameters at H = 20


@pytest.mark.parametrize('network,n', [('power_law', 6000), ('small_world', 6000), ('random', 3000)])
def test_irregular_graphs_fused_path_vs_oracle(dev, network, n):
    """Configs C2-C4 in miniature: hub rows longer than one 64-entry chunk (Barabasi-Albert), random shortcuts,
    G(n,p) with mean degree 40 - through the fused H = 256 kernel and its RK epilogues, against the CPU oracle."""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    G = graphs.make_graph(network, n, seed=2)
    L = graphs.normalized_laplacian(G)
    if network == 'power_law':
        assert np.diff(L.indptr).max() > 64                          # exercises the long-row path
    torch.manual_seed(7)
    f = ODEFunc(256, graphs.to_device(L, dev)).to(dev).eval()
    x0 = torch.rand(n, 256, generator=torch.Generator().manual_seed(8))
    t = torch.tensor([0., 0.5, 1.0])
    log = []
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), rtol=.01, atol=.001, method='dopri5', step_log=log)
        yr = ode.odeint(f, x0.to(dev), t.to(dev), method='rk4')
    Ao = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
    fo = orc.OracleODEFunc(Ao, f.wt.weight.detach().cpu(), f.wt.bias.detach().cpu())
    lo = []
    ref = orc.odeint(fo, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lo)
    assert [r[2] for r in log[:-1]] == [r[2] for r in lo]
    check_traj(y.cpu().numpy(), ref.numpy(), l1=1e-5, mx=3e-4)
    check_traj(yr.cpu().numpy(), orc.odeint(fo, x0, t, method='rk4').numpy(), l1=1e-5, mx=3e-4)


def _cora(dev):
    from ndcn_amd import CsrOperator
    import scipy.sparse as sp
    d = load_golden('dataset_cora')
    g = load_golden('operators_cora')
    n = int(g['n'])
    adj = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    feats = sp.csr_matrix((d['feat_data'], d['feat_indices'].astype(np.int64), d['feat_indptr']), shape=tuple(d['feat_shape']))
    return (adj, torch.from_numpy(feats.toarray()).to(dev), torch.from_numpy(d['labels'].astype(np.int64)).to(dev),
            torch.from_numpy(d['idx_train'].astype(np.int64)).to(dev), torch.from_numpy(d['idx_val'].astype(np.int64)).to(dev),
            torch.from_numpy(d['idx_test'].astype(np.int64)).to(dev))


def test_dgnn_resgcn_trains_on_cora(dev):
    """dgnn.py --model resGCN (dgnn.py:129-140): Linear -> ReLU -> 2 x ResBlock -> Linear with dropout .5, trained on
    the HIP SpMM (forward) / SpMM with A^T (backward).  A two-hop residual GCN on Cora lands in the high 70s."""
    from ndcn_amd.drivers import dgnn
    accs = dgnn.main(['--dataset', 'cora', '--model', 'resGCN', '-nhl', '2', '--hidden', '64', '--dropout', '0.5',
                      '--epochs', '100', '--weight_decay', '5e-4', '--alpha', '0', '--seed', '0'], data=_cora(dev), quiet=True)
    assert accs.mean() >= 0.72, accs


def test_dgnn_cora_accuracy_parity(dev):
    """Config C5 on Cora (the Pubmed feature blob is missing from the reference mount): the README command
    (README.md:64) - differential_gcn, hidden 256, T 1.2, 16 ticks, dopri5 rtol = atol = .1, no_control, alpha 0,
    100 epochs, weight decay .024 - trained on the HIP path.  README.md:67-73 reports 83.18 % +/- 0.76 over 5 runs
    (min 82.6, max 84.5) on the author's unpinned PyTorch; the CPU oracle under torch 2.10 (same pipeline, full
    autograd through the controller) reaches 81.6 % with seed 0, and 78.6 % if the step-size paths are cut.
    The HIP path must land with the former."""
    from ndcn_amd import CsrOperator
    from ndcn_amd.drivers import dgnn
    d = load_golden('dataset_cora')
    g = load_golden('operators_cora')
    n = int(g['n'])
    import scipy.sparse as sp
    adj = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    feats = sp.csr_matrix((d['feat_data'], d['feat_indices'].astype(np.int64), d['feat_indptr']), shape=tuple(d['feat_shape']))
    data = (adj, torch.from_numpy(feats.toarray()).to(dev), torch.from_numpy(d['labels'].astype(np.int64)).to(dev),
            torch.from_numpy(d['idx_train'].astype(np.int64)).to(dev), torch.from_numpy(d['idx_val'].astype(np.int64)).to(dev),
            torch.from_numpy(d['idx_test'].astype(np.int64)).to(dev))
    accs = dgnn.main(['--dataset', 'cora', '--model', 'differential_gcn', '--iter', '2', '--dropout', '0', '--hidden', '256',
                      '--T', '1.2', '--time_tick', '16', '--epochs', '100', '--weight_decay', '0.024', '--no_control',
                      '--method', 'dopri5', '--alpha', '0', '--seed', '0'], data=data, quiet=True)
    # The two values of one command are NOT two runs of one seed: like the reference (dgnn.py:128-183 builds model and optimiser
    # ahead of its --iter loop) iteration 2 keeps training the model of iteration 1 - epochs 101-200.  Measured on MI355X: seed 0 ->
    # 81.9 (iteration 1) / 83.4 % (iteration 2), seed 1 -> 84.5 / 83.8 %, seed 2 -> 83.9 / 83.5 %: mean 83.5 %, inside
    # README.md:67-73's span (83.18 +/- 0.76, min 82.6, max 84.5) and above the oracle's 81.6 % under torch 2.10.
    # Round 4: 4 commands x 8 iterations (tools/micro/cora_ab.py): mean 83.3 %, min 81.8, max 84.5, standard deviation 0.8 - the
    # spread over a model's training history.  The same seed gives the same bits: test_dgnn_same_seed_same_result below and
    # profiles/r05_dgnn_epoch.jsonl (losses, step-1 gradients and final parameters of two runs compared with torch.equal).
    assert accs.shape == (2,) and 0.805 <= accs.min() and accs.max() <= 0.855 and 0.815 <= accs.mean() <= 0.85, accs


def test_dgnn_same_seed_same_result(dev):
    """The reference on CPU is deterministic for a fixed seed; so is the HIP training path: every reduction on it runs in a fixed
    order (linear_bwd.hip: chunk partials summed in order; rk_bwd.hip / rk.hip: per-workgroup fp64 partials + a fixed-order finish;
    the SpMM with A^T is a row gather, no atomics).  Two runs of the README command (40 epochs) with seed 0: same accuracy, and the
    per-epoch log lines (losses to 4 decimals, accuracies) are identical text."""
    import io
    import contextlib
    from ndcn_amd.drivers import dgnn
    data = _cora(dev)
    outs = []
    for _ in range(2):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            accs = dgnn.main(['--dataset', 'cora', '--model', 'differential_gcn', '--iter', '1', '--dropout', '0', '--hidden', '256',
                              '--T', '1.2', '--time_tick', '16', '--epochs', '40', '--weight_decay', '0.024', '--no_control',
                              '--method', 'dopri5', '--alpha', '0', '--seed', '0'], data=data)
        lines = [l.split(' time: ')[0] for l in buf.getvalue().splitlines() if l.startswith('ITER')]
        outs.append((float(accs[0]), lines))
    assert len(outs[0][1]) == 40
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
