"""Column-sweep plan and kernel (ndcn_amd/csrc/csr_plan.hip: build_sweep_plan; csrc/spmm_sweep.hip) - the path of operators
without locality whose rows are long relative to their count (BASELINE config 2: heat_dynamics.py:89 at 10^5 nodes):
  * the plan is integer work: every array equals its numpy restatement (tests/_plan_reference.py) bit for bit;
  * S = A X by the sweep equals the row kernels bit for bit (per row the same fma chain in ascending column order), NaN / Inf
    included; the right-hand side (sweep + fused kernel on the identity operator) equals the one-launch fused kernel in every
    RK mode; both against fp64 / the oracle;
  * the library takes the plan by itself exactly where the fetch arithmetic says it pays."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _plan_reference import sweep_plan_reference
from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from ndcn_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _pull(ptr, n, dtype, dev):
    from ndcn_amd import _lib
    out = torch.empty(n, dtype=dtype, device=dev)
    if n:
        _lib.check(_lib.load().ndcn_copy_f32(out.data_ptr(), ptr, n, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _rand(n_rows, n_cols, avg, seed, empty_rows=True, long_row=0):
    rng = np.random.RandomState(seed)
    deg = rng.poisson(avg, size=n_rows)
    if empty_rows:
        deg[rng.randint(0, n_rows, size=max(1, n_rows // 40))] = 0
    if long_row:
        deg[rng.randint(0, n_rows)] = long_row
    rows = np.repeat(np.arange(n_rows), deg)
    cols = rng.randint(0, n_cols, size=rows.size)
    m = sp.csr_matrix(((rng.randn(rows.size) / 4).astype(np.float32), (rows, cols)), shape=(n_rows, n_cols))
    m.sum_duplicates()
    m.sort_indices()
    return m


def _swept(m, dev, H=256):
    from ndcn_amd import CsrOperator, _lib
    A = CsrOperator.from_scipy(m, dev)
    A.build_plans(H, flags=_lib.PLAN_FORCE_SWEEP | _lib.PLAN_NO_REC | _lib.PLAN_NO_HUB)    # (alpha / relu calls fall to the row kernels: no hub segments)
    A._plans_tried = True
    assert A.sweep is not None
    return A


def _plain(m, dev):
    from ndcn_amd import CsrOperator
    A = CsrOperator.from_scipy(m, dev)
    A._plans_tried = True                    # no plans: every row is gathered by the row kernels
    return A


def _bits(t):
    return t.contiguous().view(torch.int32)


CASES = {
    'square_3000': lambda: _rand(3000, 3000, 40, 1),
    'wide_700x5000': lambda: _rand(700, 5000, 25, 2),                  # fewer rows than waves: most slabs are empty
    'tall_9001x1500': lambda: _rand(9001, 1500, 12, 3, long_row=1400),  # ragged XCD chunks, a row longer than any staging buffer
    'two_passes': lambda: _rand(100352 + 777, 2100, 3, 4),            # 101 129 rows: two passes of 50 565 rows
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_sweep_plan_equals_the_restatement(dev, name):
    m = CASES[name]()
    A = _swept(m, dev)
    ref = sweep_plan_reference(m.indptr, m.indices, m.data, m.shape)
    v = A.view()
    assert (v.sweep_passes, v.sweep_rows_per_pass, v.sweep_rpw) == (ref['passes'], ref['rows_per_pass'], ref['rpw'])
    assert A.sweep['entries'] == ref['entries'] and A.sweep['passes'] == ref['passes']
    slab = _pull(v.sweep_slab, ref['slab'].size, torch.int32, dev).reshape(-1, 2)
    assert np.array_equal(slab, ref['slab'])
    ent = _pull(v.sweep_ent, ref['ent'].size, torch.int32, dev).view(np.uint32).reshape(-1, 2)
    assert np.array_equal(ent, ref['ent'])
    n = m.shape[0]
    assert np.array_equal(_pull(v.sweep_eye_rowptr, n + 1, torch.int32, dev), np.arange(n + 1))
    assert np.array_equal(_pull(v.sweep_eye_colidx, n, torch.int32, dev), np.arange(n))
    assert np.array_equal(_pull(v.sweep_eye_val, n, torch.float32, dev), np.ones(n, np.float32))
    # the plan decodes back to the operator: entries of slab s, row r -> (row0 + r, column, value)
    rows, cols, vals = [], [], []
    for s, (start, cnt) in enumerate(ref['slab']):
        if cnt:
            e = ent[start:start + cnt]
            p, x, sl = s // 2048, (s % 2048) // 256, s % 256
            base = p * ref['rows_per_pass']
            end = min(n, base + ref['rows_per_pass'])
            per_xcd = (end - base + 7) // 8
            rpw = (per_xcd + 255) // 256
            rows.append(base + x * per_xcd + sl * rpw + (e[:, 0] >> 24).astype(np.int64))
            cols.append((e[:, 0] & 0xffffff).astype(np.int64))
            vals.append(e[:, 1].copy().view(np.float32))
    back = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=m.shape)
    back.sort_indices()
    assert np.array_equal(back.indptr, m.indptr) and np.array_equal(back.indices, m.indices) and np.array_equal(back.data, m.data)


@pytest.mark.parametrize('name', sorted(CASES))
def test_sweep_spmm_is_the_sequential_fma_chain(dev, name):
    """S = A X: the sweep against the row kernel on the plan-free operator - torch.equal on the bits, with NaN / Inf / -0 in X."""
    from ndcn_amd import hip
    m = CASES[name]()
    A, R = _swept(m, dev), _plain(m, dev)
    g = torch.Generator().manual_seed(5)
    X = (torch.rand(m.shape[1], 256, generator=g) - 0.5).to(dev)
    got, ref = hip.spmm(A, X), hip.spmm(R, X)
    assert torch.equal(_bits(got), _bits(ref))
    exact = torch.from_numpy(m.astype(np.float64) @ X.cpu().double().numpy())
    bound = torch.from_numpy(abs(m).astype(np.float64) @ X.cpu().double().abs().numpy()) * 2e-6 + 1e-30
    assert ((got.cpu().double() - exact).abs() <= bound).all()
    Xb = X.clone()
    Xb[3, 7] = float('nan')
    Xb[11, :] = float('inf')
    Xb[17, 0] = -0.0
    Xb[min(m.shape[1] - 1, 1400), 200] = -float('inf')
    assert torch.equal(_bits(hip.spmm(A, Xb)), _bits(hip.spmm(R, Xb)))
    # twice in a row (the progress line carries the previous launch's tags) and into a caller's buffer
    out = torch.empty_like(got)
    hip.spmm(A, X, out=out)
    assert torch.equal(_bits(out), _bits(ref))
    # alpha / relu are not the sweep's business: the row kernels serve them on the same operator
    assert torch.equal(hip.spmm(A, X, alpha=-0.5, relu=True), hip.spmm(R, X, alpha=-0.5, relu=True))


@pytest.mark.parametrize('name', ['square_3000', 'tall_9001x1500'])
def test_sweep_rhs_equals_the_one_launch_fused_kernel(dev, name):
    """relu(W (A X) + b) with every RK epilogue: the sweep + the fused kernel on the identity operator over S = A X against the
    fused kernel gathering A's rows itself (plan-free operator): same S bits, same dense product, same epilogue."""
    from ndcn_amd import hip, _lib
    m = CASES[name]()
    if m.shape[0] != m.shape[1]:
        m = m[:, :m.shape[0]].tocsr() if m.shape[1] > m.shape[0] else sp.hstack([m, sp.csr_matrix((m.shape[0], m.shape[0] - m.shape[1]), dtype=np.float32)]).tocsr()
        m.sort_indices()
    n, H = m.shape[0], 256
    A, R = _swept(m, dev), _plain(m, dev)
    g = torch.Generator().manual_seed(9)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    lib = _lib.load()
    K = hip.rhs(A, X, W, b)
    assert lib.ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3 | _lib.PATH_SWEEP
    Kr = hip.rhs(R, X, W, b)
    assert lib.ndcn_debug_last_rhs_path() == _lib.PATH_FUSED2
    assert torch.equal(K, Kr)
    exact = torch.relu(torch.from_numpy(m.astype(np.float64) @ X.cpu().double().numpy()).to(dev) @ W.double().T + b.double())
    assert (K.double() - exact).abs().max() < 2e-5
    Ko = orc.odefunc_rhs(torch.from_numpy(m.toarray()), X.cpu(), W.cpu(), b.cpu()) if n <= 3000 else None       # the oracle's dense operator
    if Ko is not None:
        assert (K.cpu() - Ko).abs().max() < 1e-5
    for npv in range(6):
        c = cs[:npv] + [cs[5]]
        K1, y1 = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], c)
        K2, y2 = hip.rhs_rk(R, X, W, b, 'combine', y0, ks[:npv], c)
        assert torch.equal(K1, K2) and torch.equal(y1, y2), npv
    for st in range(4):
        K1, y1 = hip.rhs_rk(A, X, W, b, 'rk4', y0, ks[:st], [np.float32(0.37)])
        K2, y2 = hip.rhs_rk(R, X, W, b, 'rk4', y0, ks[:st], [np.float32(0.37)])
        assert torch.equal(K1, K2) and torch.equal(y1, y2), st
    Xbad = X.clone()
    Xbad[7, 5] = float('inf')
    for Xin in (X, Xbad):
        K1, (s1, b1) = hip.rhs_rk(A, Xin, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
        K2, (s2, b2) = hip.rhs_rk(R, Xin, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
        assert b1 == b2 == (0.0 if Xin is X else 1.0)
        if Xin is X:
            assert torch.equal(K1, K2) and abs(s1 - s2) <= 1e-12 * abs(s1)
    # no_control (relu(A X), dgnn's right-hand side): the sweep, then the ReLU / the stage algebra as streaming passes - against the
    # row gather with the algebra in its epilogue (same S bits, same op order per element)
    assert torch.equal(hip.rhs(A, X, W, b, no_control=True), hip.rhs(R, X, W, b, no_control=True))
    Xn = X.clone()
    Xn[5, 9] = float('nan')
    assert torch.equal(_bits(hip.rhs(A, Xn, W, b, no_control=True)), _bits(hip.rhs(R, Xn, W, b, no_control=True)))
    for npv in (0, 3, 5):
        c = cs[:npv] + [cs[5]]
        K1, y1 = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], c, no_control=True)
        K2, y2 = hip.rhs_rk(R, X, W, b, 'combine', y0, ks[:npv], c, no_control=True)
        assert torch.equal(K1, K2) and torch.equal(y1, y2), npv
    for st in range(4):
        K1, y1 = hip.rhs_rk(A, X, W, b, 'rk4', y0, ks[:st], [np.float32(0.37)], no_control=True)
        K2, y2 = hip.rhs_rk(R, X, W, b, 'rk4', y0, ks[:st], [np.float32(0.37)], no_control=True)
        assert torch.equal(K1, K2) and torch.equal(y1, y2), st
    K1, (s1, b1) = hip.rhs_rk(A, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3, no_control=True)
    K2, (s2, b2) = hip.rhs_rk(R, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3, no_control=True)
    assert torch.equal(K1, K2) and b1 == b2 == 0.0 and abs(s1 - s2) <= 1e-12 * abs(s1)


def test_training_through_a_swept_operator(dev):
    """loss.backward() through dopri5 and rk4 on an operator with the column-sweep plan: the forward evaluations, the S = A X the
    weight gradient re-forms and g_X = A^T g_S (the transposed operator gets a plan of its own) all run the sweep - gradients
    against the plan-free operator."""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    n, H = 9000, 256
    m = _rand(n, n, 40, 6)
    op = graphs.normalized_laplacian(sp.csr_matrix(abs(m) + abs(m).T))
    t = torch.linspace(0., 1., 3).to(dev)
    x0 = torch.rand(n, H, generator=torch.Generator().manual_seed(2)).to(dev)
    w = torch.randn(3, n, H, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for name, mk in (('swept', lambda: graphs.to_device(op, dev)), ('plain', lambda: _plain(op, dev))):
        for method, kw in (('dopri5', dict(rtol=1e-2, atol=1e-3)), ('rk4', {})):
            torch.manual_seed(0)
            A = mk()
            f = ODEFunc(H, A).to(dev)
            x = x0.clone().requires_grad_(True)
            y = ode.odeint(f, x, t, method=method, **kw)
            (y * w).sum().backward()
            if name == 'swept':
                assert A.sweep is not None and A.transpose().sweep is not None
            res[name, method] = (y.detach(), x.grad, f.wt.weight.grad, f.wt.bias.grad)
    for method in ('dopri5', 'rk4'):
        for got, ref in zip(res['swept', method], res['plain', method]):
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7, method


def test_sweep_is_taken_where_the_fetch_arithmetic_says_it_pays(dev):
    """ndcn_csr_create's own decision (include/ndcn_hip.h): n_cols >= 8192 and nnz >= 1.5 * passes * 8 * n_cols, no group-record
    plan, a whole operator - and odeint on such an operator runs its solver through the sweep."""
    from ndcn_amd import CsrOperator, graphs, _lib
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    dense_enough = _rand(9000, 9000, 40, 6)
    A = CsrOperator.from_scipy(dense_enough, dev).ensure_plans(256)
    assert A.sweep is not None and A.sweep['passes'] == 1 and A.sweep['rows_per_wave'] == 5
    for m in (_rand(9000, 9000, 10, 7),                                                        # too few entries per column
              _rand(4000, 4000, 40, 8),                                                        # the panel fits the L2s anyway
              graphs.normalized_laplacian(graphs.grid_8_neighbor(100)),                        # has a group-record plan
              graphs.normalized_laplacian(graphs.make_graph('small_world', 10000, seed=3))):
        m = m.tocsr()
        m.sort_indices()
        assert CsrOperator.from_scipy(m, dev).ensure_plans(256).sweep is None
    assert CsrOperator.from_scipy(dense_enough, dev).ensure_plans(64).sweep is None          # plans serve H = 256
    sharded = CsrOperator.from_scipy(dense_enough, dev)
    sharded.n_halo = 10
    assert sharded.ensure_plans(256).sweep is None
    # a solve on the swept operator: dopri5 and rk4 through the device-resident solver vs the plan-free operator
    n, H = 9000, 256
    torch.manual_seed(0)
    op = graphs.normalized_laplacian(sp.csr_matrix(abs(dense_enough) + abs(dense_enough).T))
    f = ODEFunc(H, graphs.to_device(op, dev)).to(dev)
    x0 = torch.rand(n, H).to(dev)
    t = torch.linspace(0., 1., 4).to(dev)
    with torch.no_grad():
        for method, kw in (('dopri5', dict(rtol=1e-2, atol=1e-3)), ('rk4', {})):
            y = ode.odeint(f, x0, t, method=method, **kw)
            assert _lib.load().ndcn_debug_last_rhs_path() & _lib.PATH_SWEEP
            f2 = ODEFunc(H, _plain(op, dev)).to(dev)
            f2.load_state_dict(f.state_dict())
            y2 = ode.odeint(f2, x0, t, method=method, **kw)
            assert torch.allclose(y, y2, rtol=1e-5, atol=1e-6), method


def test_sweep_at_the_size_of_config_2(dev):
    """BASELINE config 2 as bench.py builds it (10^5-node G(n,p), mean degree 40): the plan is automatic, one pass of 49 rows per
    wave; S bit-equal to the row kernel; sampled rows of the right-hand side against fp64."""
    from ndcn_amd import hip, CsrOperator, graphs, _lib
    n, H = 100000, 256
    m = graphs.normalized_laplacian(graphs.make_graph('random', n, seed=0)).tocsr()
    m.sort_indices()
    A = CsrOperator.from_scipy(m, dev).ensure_plans(H)
    assert A.sweep is not None and (A.sweep['passes'], A.sweep['rows_per_wave']) == (1, 49)
    R = _plain(m, dev)
    g = torch.Generator().manual_seed(1)
    X = torch.rand(n, H, generator=g).to(dev)
    S = hip.spmm(A, X)
    assert torch.equal(_bits(S), _bits(hip.spmm(R, X)))
    # many launches back to back (the progress lines carry one launch's tags into the next; the tag counter advances on the device):
    # every result is the same bits, and the launches keep their pace (a wave that waits in vain would show as milliseconds)
    out = torch.empty_like(S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(4):
        e0.record()
        for _ in range(100):
            hip.spmm(A, X, out=out)
        e1.record()
        torch.cuda.synchronize()
        assert torch.equal(_bits(out), _bits(S)), rep
        assert e0.elapsed_time(e1) / 100 < 1.0, e0.elapsed_time(e1) / 100            # 0.22 ms on MI355X; the row gather takes 0.59
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    K = hip.rhs(A, X, W, b)
    assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3 | _lib.PATH_SWEEP
    rows = np.random.RandomState(0).choice(n, 64, replace=False)
    Xd = X.cpu().double().numpy()
    Sd = m[rows].astype(np.float64) @ Xd
    exact = np.maximum(Sd @ W.cpu().double().numpy().T + b.cpu().double().numpy(), 0.0)
    mag = (abs(m[rows]).astype(np.float64) @ np.abs(Xd)) @ np.abs(W.cpu().double().numpy()).T + np.abs(b.cpu().double().numpy())
    assert (np.abs(K[torch.from_numpy(rows).to(dev)].cpu().double().numpy() - exact) <= 2e-6 * mag).all()
