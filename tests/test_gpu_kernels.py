"""Kernel-level parity on a real MI355X: every C-ABI entry point against the CPU oracle / golden fixtures.
All calls go through libndcn_hip.so (ndcn_amd.ops.hip)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, load_golden
from oracle import ndcn_oracle as orc
from _oracle_ops import OracleOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from ndcn_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.asarray(a))


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


def rand_csr(n_rows, n_cols, avg, seed, hubs=0):
    rng = np.random.RandomState(seed)
    deg = rng.poisson(avg, size=n_rows)
    deg[rng.randint(0, n_rows, size=max(1, n_rows // 50))] = 0          # some empty rows
    for h in range(hubs):
        deg[rng.randint(0, n_rows)] = min(n_cols, 3000 + 500 * h)        # rows longer than the LDS stage
    rows = np.repeat(np.arange(n_rows), deg)
    cols = rng.randint(0, n_cols, size=rows.size)
    m = sp.csr_matrix((rng.randn(rows.size).astype(np.float32), (rows, cols)), shape=(n_rows, n_cols))
    m.sum_duplicates()
    m.sort_indices()
    return m


# ------------------------------------------------------------------------------------------- SpMM
@pytest.mark.parametrize('H', [1, 2, 3, 4, 8, 20, 32, 64, 100, 128, 256, 260, 512])
def test_spmm_vs_fp64(dev, H):
    from ndcn_amd import hip, CsrOperator
    m = rand_csr(3001, 3001, 9, seed=H, hubs=2)
    X = torch.randn(3001, H)
    Y = hip.spmm(CsrOperator.from_scipy(m, dev), X.to(dev)).cpu().numpy()
    ref = orc.spmm_f64(m.indptr, m.indices, m.data, X.numpy())
    scale = np.abs(m).dot(np.abs(X.numpy()).astype(np.float64))
    assert np.all(np.abs(Y - ref) <= 2e-6 * scale + 1e-6)


def test_spmm_alpha_relu_rect_and_empty(dev):
    from ndcn_amd import hip, CsrOperator
    m = rand_csr(517, 1200, 5, seed=3)
    X = torch.randn(1200, 36)
    A = CsrOperator.from_scipy(m, dev)
    Y = hip.spmm(A, X.to(dev), alpha=-2.5, relu=True).cpu().numpy()
    ref = np.maximum(-2.5 * orc.spmm_f64(m.indptr, m.indices, m.data, X.numpy()), 0)
    assert np.abs(Y - ref).max() < 1e-4
    # N x 1 and 1-D panels (the truth dynamics' layout)
    v = torch.randn(1200)
    y1 = hip.spmm(A, v.to(dev)).cpu().numpy()
    assert np.abs(y1 - orc.spmm_f64(m.indptr, m.indices, m.data, v.numpy().reshape(-1, 1)).ravel()).max() < 1e-4
    # empty operator / zero rows
    E = CsrOperator.from_scipy(sp.csr_matrix((64, 64), dtype=np.float32), dev)
    assert float(hip.spmm(E, torch.randn(64, 8).to(dev)).abs().max()) == 0.0
    Z = CsrOperator.from_scipy(sp.csr_matrix((0, 64), dtype=np.float32), dev)
    assert hip.spmm(Z, torch.randn(64, 8).to(dev)).shape == (0, 8)


def test_spmm_halo_split_equals_whole(dev):
    from ndcn_amd import hip, CsrOperator
    m = rand_csr(800, 2000, 7, seed=11)
    X = torch.randn(2000, 64).to(dev)
    A = CsrOperator.from_scipy(m, dev)
    whole = hip.spmm(A, X)
    split = hip.spmm(A, X[:1100].contiguous(), X_halo=X[1100:].contiguous())
    assert torch.equal(whole, split)


def test_spmm_dense_and_coo_inputs_match_reference_layouts(dev):
    from ndcn_amd import hip
    d = load_golden('rhs_grid400_H20_no_control_coo')
    dense = orc.dense_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    coo = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    x = T(d['x']).to(dev)
    y_dense = hip.spmm(dense.to(dev), x, relu=True).cpu().numpy()
    y_coo = hip.spmm(coo.to(dev), x, relu=True).cpu().numpy()
    assert np.array_equal(y_dense, y_coo)
    assert np.abs(y_coo - d['out']).max() < 1e-5


def _no_plan(A):
    A._plans_tried = True                                         # keep the operator plan-free (direct-gather kernels)
    return A


@pytest.mark.parametrize('shape', [(8, 32, 1), (16, 40, 2), (8, 48, 2)])
@pytest.mark.parametrize('side', [45, 64])
def test_spmm_group_record_kernel(dev, shape, side):
    """H = 256 with a group-record plan (spmm_rec.hip): staged groups, groups the record cannot hold (direct gather
    inside the kernel), ragged last group, lattice patches sticking out of the grid, halo panel, alpha / relu - all
    must equal the plan-free kernel bit for bit (same summation order)."""
    from ndcn_amd import hip, CsrOperator, graphs
    n = side * side
    grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    rnd = rand_csr(n, n, 9, seed=5, hubs=2)                                          # no sharing, long rows -> direct
    mixed = sp.vstack([grid[:1000], rnd[1000:]]).tocsr()
    X = torch.randn(n, 256).to(dev)
    for m, hinted in ((grid, True), (grid, False), (mixed, False), (rnd, False)):
        m.sort_indices()
        plain = _no_plan(CsrOperator.from_scipy(m, dev))
        ref = hip.spmm(plain, X)
        A = _no_plan(CsrOperator.from_scipy(m, dev))
        if hinted:
            A.group_order = torch.as_tensor(A.detect_stencil_order(), dtype=torch.int32).to(dev)
        staged, _ = A.build_rec_plan(*shape)
        if m is grid and (hinted or shape[0] == 8):
            assert staged == 1.0
        if m is rnd:
            assert staged < 0.05
        assert A.view().rec_groups > 0
        assert torch.equal(hip.spmm(A, X), ref)
        assert torch.equal(hip.spmm(A, X, alpha=-1.5, relu=True), hip.spmm(plain, X, alpha=-1.5, relu=True))
        got = hip.spmm(A, X[:1200].contiguous(), X_halo=X[1200:].contiguous())      # halo split
        assert torch.equal(got, ref)
    ref64 = orc.spmm_f64(grid.indptr, grid.indices, grid.data, X.cpu().numpy())
    assert np.abs(hip.spmm(CsrOperator.from_scipy(grid, dev), X).cpu().numpy() - ref64).max() < 1e-4


def test_small_world_operator_gets_the_ring_plan_automatically(dev):
    """Newman-Watts-Strogatz graphs (config C4, gene_dynamics.py:99,103): 8 consecutive rows share their ring neighbours
    (12 columns) and add ~2 random shortcut endpoints each - too many for the 32-column record, fine for {8, 48, 2}:
    ensure_plans attaches it, the SpMM and the no_control RHS + RK epilogue run the group-record kernel, bit-equal to the
    plan-free row kernel."""
    from ndcn_amd import hip, CsrOperator, graphs
    n, H = 30000, 256
    m = graphs.normalized_laplacian(graphs.make_graph('small_world', n, seed=0)).tocsr()
    A = CsrOperator.from_scipy(m, dev)
    A.ensure_plans(H)
    assert A.rec is not None and (A.rec['rows'], A.rec['cap'], A.rec['kib']) == (8, 48, 2) and A.rec['staged'] > 0.9
    P = _no_plan(CsrOperator.from_scipy(m, dev))
    g = torch.Generator().manual_seed(1)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(2)]
    assert torch.equal(hip.spmm(A, X), hip.spmm(P, X))
    cs = [np.float32(0.2), np.float32(-0.1), np.float32(0.3)]
    K1, y1 = hip.rhs_rk(A, X, None, None, 'combine', y0, ks, cs, no_control=True)
    K2, y2 = hip.rhs_rk(P, X, None, None, 'combine', y0, ks, cs, no_control=True)
    assert torch.equal(K1, K2) and torch.equal(y1, y2)
    # an Erdos-Renyi graph of the same density shares nothing: no plan
    B = CsrOperator.from_scipy(graphs.normalized_laplacian(graphs.make_graph('random', 5000, seed=0, mean_degree=7)).tocsr(), dev)
    B.ensure_plans(H)
    assert B.rec is None


def test_lattice_operator_gets_patch_plan_automatically(dev):
    """A lattice operator handed over as a plain tensor (the reference's way) is recognised and gets the 16-row patch
    plan; a random graph gets none."""
    from ndcn_amd import hip, CsrOperator, graphs
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(50))
    A = CsrOperator.from_torch(torch.from_numpy(L.toarray()).to(dev))
    A.ensure_plans(256)
    assert A.rec is not None and A.rec['rows'] == 16 and A.rec['staged'] == 1.0 and A.rec['loads_per_row'] < 2.6
    B = CsrOperator.from_scipy(rand_csr(2000, 2000, 9, seed=1), dev)
    B.ensure_plans(256)
    assert B.rec is None
    X = torch.randn(2500, 256).to(dev)
    assert torch.equal(hip.spmm(A, X), hip.spmm(_no_plan(CsrOperator.from_scipy(L, dev)), X))


@pytest.mark.parametrize('shape', [(8, 32, 1), (16, 40, 2), (8, 48, 2), None])
def test_no_control_rhs_rk_epilogue_in_group_record_kernel(dev, shape):
    """relu(A X) with the stage algebra in the SpMM's epilogue (the no_control RHS of the dgnn README command):
    COMBINE with 0..5 earlier stages, ERROR, RK4 stages 0..3 - bit-identical to SpMM + the separate stage kernels.
    shape None: no plan - the row SpMM kernel carries the epilogue (any graph)."""
    from ndcn_amd import hip, CsrOperator, graphs
    side, H = 41, 256
    n = side * side
    grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    rnd = rand_csr(n, n, 7, seed=9)
    for m in (grid, sp.vstack([grid[:800], rnd[800:]]).tocsr()):
        m.sort_indices()
        A = _no_plan(CsrOperator.from_scipy(m, dev))
        if shape is not None:
            A.group_order = torch.as_tensor(CsrOperator.from_scipy(grid, dev).detect_stencil_order(), dtype=torch.int32).to(dev)
            A.build_rec_plan(*shape)
        P = _no_plan(CsrOperator.from_scipy(m, dev))
        g = torch.Generator().manual_seed(2)
        X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
        ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
        cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
        K_ref = hip.rhs(P, X, None, None, no_control=True)
        assert torch.equal(hip.rhs(A, X, None, None, no_control=True), K_ref)
        for npv in range(6):
            K, yn = hip.rhs_rk(A, X, None, None, 'combine', y0, ks[:npv], cs[:npv] + [cs[5]], no_control=True)
            assert torch.equal(K, K_ref)
            assert torch.equal(yn, hip.combine(y0, ks[:npv] + [K_ref], cs[:npv] + [cs[5]]))
        for npv in (5, 3, 0) * 3:                                  # repeated: a premature read of a panel is a race
            K, (s1, b1) = hip.rhs_rk(A, X, None, None, 'error', y0, ks[:npv], cs[:npv] + [cs[5]], rtol=1e-2, atol=1e-3,
                                     no_control=True)
            s2, b2 = hip.error(y0, X, ks[:npv] + [K_ref], cs[:npv] + [cs[5]], 1e-2, 1e-3)
            # (the stand-alone kernel sums a panel of this size in ATen's float32 order, the epilogue in fp64)
            assert torch.equal(K, K_ref) and abs(s1 - s2) <= 1e-5 * abs(s2) and b1 == b2 == 0.0
        Xbad = X.clone()
        Xbad[5, 7] = float('inf')
        _, (_, bad) = hip.rhs_rk(A, Xbad, None, None, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3, no_control=True)
        assert bad == 1.0
        dt = np.float32(0.37)
        for st in range(4):
            K, yn = hip.rhs_rk(A, X, None, None, 'rk4', y0, ks[:st], [dt], no_control=True)
            kk = ks[:st] + [K_ref]
            want = hip.fixed_stage(2 + st, y0, *kk, dt=dt)
            assert torch.equal(K, K_ref) and torch.equal(yn, want)


@pytest.mark.parametrize('side', [41, 64])
def test_fused3_equals_fused2(dev, side):
    """The whole ODEFunc + RK epilogue on an operator with the 16-row group-record plan (rhs_fused3.hip: LDS-DMA staging,
    32-row S tiles) against the same operator without a plan (rhs_fused2.hip: register gather, 64-row tiles): staged
    groups in lattice-patch order and in row order, groups the record cannot hold (gathered directly inside the kernel),
    a ragged last group, an odd number of groups per workgroup, a halo panel - same fold order, same k order, same
    products: bit-equal K, y_next; error sums to fp64 rounding."""
    from ndcn_amd import hip, CsrOperator, graphs
    H = 256
    n = side * side
    grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    rnd = rand_csr(n, n, 9, seed=5, hubs=2)
    mixed = sp.vstack([grid[:1000], rnd[1000:]]).tocsr()
    g = torch.Generator().manual_seed(4)
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    for m, hinted in ((grid, True), (grid, False), (mixed, False)):
        m.sort_indices()
        P = _no_plan(CsrOperator.from_scipy(m, dev))
        A = _no_plan(CsrOperator.from_scipy(m, dev))
        if hinted:
            A.group_order = torch.as_tensor(A.detect_stencil_order(), dtype=torch.int32).to(dev)
        staged, _ = A.build_rec_plan(16, 40, 2)
        assert A.view().rec_groups > 0 and (staged == 1.0 if hinted else staged < 1.0 or m is grid)
        from ndcn_amd import _lib
        K_ref = hip.rhs(P, X, W, b)
        assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED2
        assert torch.equal(hip.rhs(A, X, W, b), K_ref)
        assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
        got = hip.rhs(A, X[:1200].contiguous(), W, b, X_halo=X[1200:].contiguous())
        assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3 | _lib.PATH_HALO
        assert torch.equal(got, K_ref)
        # the HALO variants of the epilogue modes: own rows [0, 1200), the rest of X as the halo panel
        sub = m[:1200]
        Ah = _no_plan(CsrOperator.from_scipy(sub, dev))
        Ah.lattice_hint = (0, 1200)
        if hinted:
            Ah.group_order = torch.as_tensor(Ah.detect_stencil_order(), dtype=torch.int32).to(dev)
        Ah.build_rec_plan(16, 40, 2)
        Xo, Xh = X[:1200].contiguous(), X[1200:].contiguous()
        Kh, ynh = hip.rhs_rk(Ah, Xo, W, b, 'combine', y0[:1200], [k[:1200] for k in ks[:3]], cs[:3] + [cs[5]], X_halo=Xh)
        assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3 | _lib.PATH_HALO
        Kf, ynf = hip.rhs_rk(P, X, W, b, 'combine', y0, ks[:3], cs[:3] + [cs[5]])
        assert torch.equal(Kh, Kf[:1200]) and torch.equal(ynh, ynf[:1200])
        _, (se, be) = hip.rhs_rk(Ah, Xo, W, b, 'error', y0[:1200], [k[:1200] for k in ks], cs, rtol=1e-2, atol=1e-3, X_halo=Xh)
        Ph = _no_plan(CsrOperator.from_scipy(sub, dev))
        _, (se2, be2) = hip.rhs_rk(Ph, Xo, W, b, 'error', y0[:1200], [k[:1200] for k in ks], cs, rtol=1e-2, atol=1e-3, X_halo=Xh)
        assert abs(se - se2) <= 1e-9 * abs(se2) and be == be2 == 0.0
        for npv in range(6):
            K, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], cs[:npv] + [cs[5]])
            K2, yn2 = hip.rhs_rk(P, X, W, b, 'combine', y0, ks[:npv], cs[:npv] + [cs[5]])
            assert torch.equal(K, K_ref) and torch.equal(K2, K_ref) and torch.equal(yn, yn2)
        for _ in range(3):                                          # repeated: a premature read of a panel is a race
            K, (s1, b1) = hip.rhs_rk(A, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
            _, (s2, b2) = hip.rhs_rk(P, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
            assert torch.equal(K, K_ref) and abs(s1 - s2) <= 1e-9 * abs(s2) and b1 == b2 == 0.0
        Xbad = X.clone()
        Xbad[5, 7] = float('inf')
        _, (_, bad) = hip.rhs_rk(A, Xbad, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
        assert bad == 1.0
        dt = np.float32(0.37)
        for st in range(4):
            K, yn = hip.rhs_rk(A, X, W, b, 'rk4', y0, ks[:st], [dt])
            assert torch.equal(K, K_ref) and torch.equal(yn, hip.fixed_stage(2 + st, y0, *(ks[:st] + [K_ref]), dt=dt))
    ref64 = np.maximum(orc.spmm_f64(grid.indptr, grid.indices, grid.data, X.cpu().numpy()) @ W.cpu().double().numpy().T
                       + b.cpu().double().numpy(), 0)
    A = CsrOperator.from_scipy(grid, dev)
    A.ensure_plans(H)
    assert A.rec is not None and np.abs(hip.rhs(A, X, W, b).cpu().numpy() - ref64).max() < 2e-5


def test_group_record_kernels_beyond_two_million_rows(dev):
    """Row offsets row << 10 pass 2^31 bytes at 2^21 rows: the group-record kernels address row-local panels as a 64-bit
    base + a 32-bit per-lane byte offset that must be read as UNSIGNED.  1500 x 1500 lattice (2.25 M rows, 2.3 GB panels):
    rhs_fused3 / spmm_rec against the plan-free kernels (buffer descriptors) on the whole panel, last rows against fp64."""
    from ndcn_amd import hip, CsrOperator, graphs, _lib
    side, H = 1500, 256
    n = side * side
    assert n > (1 << 21)
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = CsrOperator.from_scipy(L, dev)
    A.ensure_plans(H)
    assert A.rec is not None and A.rec['rows'] == 16
    P = _no_plan(CsrOperator.from_scipy(L, dev))
    g = torch.Generator().manual_seed(6)
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    X = torch.rand(n, H, device=dev)
    y0 = torch.rand(n, H, device=dev)
    k1 = torch.randn(n, H, device=dev)
    cs = [np.float32(0.11), np.float32(0.19)]
    assert torch.equal(hip.spmm(A, X), hip.spmm(P, X))
    K, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, [k1], cs)
    assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
    K2, yn2 = hip.rhs_rk(P, X, W, b, 'combine', y0, [k1], cs)
    assert torch.equal(K, K2) and torch.equal(yn, yn2)
    Kn, ynn = hip.rhs_rk(A, X, None, None, 'combine', y0, [k1], cs, no_control=True)          # spmm_rec epilogue
    Kn2, ynn2 = hip.rhs_rk(P, X, None, None, 'combine', y0, [k1], cs, no_control=True)
    assert torch.equal(Kn, Kn2) and torch.equal(ynn, ynn2)
    rows = np.r_[n - 64:n]
    sub = L[rows]
    S = orc.spmm_f64(sub.indptr, sub.indices, sub.data, X.cpu().numpy())
    ref = np.maximum(S @ W.cpu().double().numpy().T + b.cpu().double().numpy(), 0)
    assert np.abs(K[n - 64:].cpu().double().numpy() - ref).max() < 2e-5


def test_gather_rows(dev):
    from ndcn_amd import hip
    X = torch.randn(1000, 20).to(dev)
    idx = torch.randint(0, 1000, (333,), dtype=torch.int32).to(dev)
    assert torch.equal(hip.gather_rows(X, idx), X[idx.long()])
    X = torch.randn(1000, 256).to(dev)
    assert torch.equal(hip.gather_rows(X, idx), X[idx.long()])


# ------------------------------------------------------------------------------------------- Linear / RHS
@pytest.mark.parametrize('n,Hi,Ho', [(400, 20, 20), (1000, 256, 256), (777, 1, 20), (777, 20, 1), (130, 64, 48),
                                      (65, 100, 200), (4096, 128, 128), (33, 256, 7), (50, 17, 33)])
def test_linear_vs_torch_fp32(dev, n, Hi, Ho):
    from ndcn_amd import hip
    torch.manual_seed(n + Hi)
    S, W, b = torch.randn(n, Hi), torch.randn(Ho, Hi) / Hi ** 0.5, torch.randn(Ho)
    ref = torch.nn.functional.linear(S.double(), W.double(), b.double())
    for relu in (False, True):
        for bias in (b, None):
            y = hip.linear(S.to(dev), W.to(dev), None if bias is None else bias.to(dev), relu=relu).cpu().double()
            r = ref - (0 if bias is not None else b.double())
            r = torch.relu(r) if relu else r
            assert (y - r).abs().max() < 2e-5 * max(1.0, float(r.abs().max()))


def test_linear_is_an_exact_fp32_fma_chain(dev):
    # v_mfma_f32_32x32x2_f32 accumulates k in order with one rounding per product-add: compare bitwise
    from ndcn_amd import hip
    torch.manual_seed(1)
    S, W = torch.randn(64, 64), torch.randn(64, 64)
    y = hip.linear(S.to(dev), W.to(dev)).cpu().numpy()
    ref = np.zeros((64, 64), dtype=np.float32)
    Sn, Wn = S.numpy(), W.numpy()
    for k in range(64):
        ref = (Sn[:, k:k + 1].astype(np.float64) * Wn[:, k][None, :].astype(np.float64) + ref.astype(np.float64)).astype(np.float32)
    assert np.array_equal(y, ref)
    # transposition detector: an asymmetric weight
    W2 = torch.zeros(64, 64); W2[3, 5] = 1.0
    out = hip.linear(torch.eye(64).to(dev), W2.to(dev)).cpu()
    assert out[5, 3] == 1.0 and out.sum() == 1.0


@pytest.mark.parametrize('name', names('rhs_*.npz'))
def test_rhs_golden(dev, name):
    from ndcn_amd import hip, CsrOperator
    d = load_golden(name)
    A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)
    out = hip.rhs(A, T(d['x']).to(dev), T(d['W']).to(dev), T(d['b']).to(dev),
                  no_graph='no_graph' in name, no_control='no_control' in name).cpu().numpy()
    assert np.abs(out - d['out']).max() <= 2e-5


def test_rhs_module_matches_reference_semantics(dev):
    # ODEFunc module incl. state_dict round trip through the reference's key names
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden('rhs_grid400_H20_default_coo')
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape']).to(dev)
    f = ODEFunc(20, A).to(dev)
    f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
    with torch.no_grad():
        out = f(torch.tensor(0.0), T(d['x']).to(dev)).cpu().numpy()
    assert np.abs(out - d['out']).max() <= 2e-5


def _shard(m, lo, hi):
    """Rows [lo, hi) of a square scipy CSR with columns remapped to [own | halo] the way sharding.HaloPlan does."""
    blk = m[lo:hi].tocoo()
    own = (blk.col >= lo) & (blk.col < hi)
    halo_cols = np.unique(blk.col[~own])
    col = np.where(own, blk.col - lo, (hi - lo) + np.searchsorted(halo_cols, blk.col))
    out = sp.csr_matrix((blk.data, (blk.row, col)), shape=(hi - lo, hi - lo + halo_cols.size))
    out.sort_indices()
    return out, halo_cols


@pytest.mark.parametrize('side,cut', [(40, 800), (33, 500)])
def test_fused_rhs_rk_with_halo_panel_equals_unsharded(dev, side, cut):
    """Every compiled (mode, stage-count) variant of the fused RHS kernel with the operand split into an own panel
    and a halo panel - the multi-GPU form - against the one-panel launch on the whole graph (what the N>1 bench runs
    but a 1-GPU box cannot: both shards are evaluated here on one device)."""
    from ndcn_amd import hip, CsrOperator, graphs
    H, n = 256, side * side
    m = graphs.make_operator(graphs.grid_8_neighbor(side), 'norm_lap').tocsr()
    g = torch.Generator().manual_seed(side)
    X = torch.rand(n, H, generator=g)
    y0 = torch.rand(n, H, generator=g)
    ks = [torch.randn(n, H, generator=g) for _ in range(5)]
    W = (torch.rand(H, H, generator=g) - 0.5) / 8
    b = (torch.rand(H, generator=g) - 0.5) / 8
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    A = CsrOperator.from_scipy(m, dev)
    d = lambda t: t.to(dev)
    Xd, y0d, ksd, Wd, bd = d(X), d(y0), [d(k) for k in ks], d(W), d(b)
    shards = []
    for lo, hi in ((0, cut), (cut, n)):
        blk, halo_cols = _shard(m, lo, hi)
        shards.append((lo, hi, CsrOperator.from_scipy(blk, dev), Xd[torch.from_numpy(halo_cols).to(dev)].contiguous()))
    full = hip.rhs(A, Xd, Wd, bd)
    for lo, hi, As, Xh in shards:
        got = hip.rhs(As, Xd[lo:hi].contiguous(), Wd, bd, X_halo=Xh)
        assert torch.allclose(got, full[lo:hi], rtol=1e-5, atol=1e-6)
    for n_prev in range(6):
        c = cs[:n_prev] + [cs[5]]
        K, ynext = hip.rhs_rk(A, Xd, Wd, bd, 'combine', y0d, ksd[:n_prev], c)
        for lo, hi, As, Xh in shards:
            Ks, ys = hip.rhs_rk(As, Xd[lo:hi].contiguous(), Wd, bd, 'combine', y0d[lo:hi].contiguous(),
                                [k[lo:hi].contiguous() for k in ksd[:n_prev]], c, X_halo=Xh)
            assert torch.allclose(Ks, K[lo:hi], rtol=1e-5, atol=1e-6)
            assert torch.allclose(ys, ynext[lo:hi], rtol=1e-5, atol=1e-6)
    K, (ss, bad) = hip.rhs_rk(A, Xd, Wd, bd, 'error', y0d, ksd, cs, rtol=1e-2, atol=1e-3)
    tot = 0.0
    for lo, hi, As, Xh in shards:
        Ks, (s1, b1) = hip.rhs_rk(As, Xd[lo:hi].contiguous(), Wd, bd, 'error', y0d[lo:hi].contiguous(),
                                  [k[lo:hi].contiguous() for k in ksd], cs, rtol=1e-2, atol=1e-3, X_halo=Xh)
        assert torch.allclose(Ks, K[lo:hi], rtol=1e-5, atol=1e-6)
        assert float(b1) == 0.0
        tot += float(s1)
    assert abs(tot - float(ss)) <= 1e-6 * abs(float(ss)) and float(bad) == 0.0


@pytest.mark.parametrize('graph', ['small_world', 'power_law', 'grid'])
@pytest.mark.parametrize('no_control', [False, True])
def test_two_phase_evaluation_equals_the_one_launch_form(dev, graph, no_control):
    """Shards whose halo columns are scattered evaluate A X in two phases (sharding.HaloPlan.two_phase): S = A_own X while
    the exchange is in flight, then the fused launch on [I | A_halo] over [S | X_halo] - with the error record formed
    from y1 passed explicitly (the launch's X is the partial sum).  Same fma sequence per row as the one-launch form:
    equal K / y_next, error sums to fp64 rounding, the non-finite count taken from y1."""
    from ndcn_amd import hip, CsrOperator, graphs
    H = 256
    if graph == 'grid':
        m = graphs.normalized_laplacian(graphs.grid_8_neighbor(50))
        lo, hi = 0, 1250
    else:
        m = graphs.normalized_laplacian(graphs.make_graph(graph, 3000, seed=3)).tocsr()
        lo, hi = (0, 1500) if graph == 'small_world' else (1500, 3000)
    n = m.shape[0]
    blk, halo_cols = _shard(m, lo, hi)
    n_own, n_halo = hi - lo, halo_cols.size
    own = blk[:, :n_own].tocsr()
    second = sp.hstack([sp.identity(n_own, dtype=np.float32, format='csr'), blk[:, n_own:]], format='csr')
    second.sort_indices()
    A1, A_own, A2 = (CsrOperator.from_scipy(x, dev) for x in (blk, own, second))
    g = torch.Generator().manual_seed(11)
    X = torch.rand(n, H, generator=g).to(dev)
    Xo, Xh = X[lo:hi].contiguous(), X[torch.from_numpy(halo_cols).to(dev)].contiguous()
    y0 = torch.rand(n_own, H, generator=g).to(dev)
    ks = [torch.randn(n_own, H, generator=g).to(dev) for _ in range(5)]
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    kw = dict(no_control=no_control)
    S = hip.spmm(A_own, Xo)
    assert torch.equal(hip.rhs(A2, S, W, b, X_halo=Xh, **kw), hip.rhs(A1, Xo, W, b, X_halo=Xh, **kw))
    for npv in (0, 2, 5):
        c = cs[:npv] + [cs[5]]
        K1, y1_ = hip.rhs_rk(A1, Xo, W, b, 'combine', y0, ks[:npv], c, X_halo=Xh, **kw)
        K2, y2_ = hip.rhs_rk(A2, S, W, b, 'combine', y0, ks[:npv], c, X_halo=Xh, **kw)
        assert torch.equal(K1, K2) and torch.equal(y1_, y2_)
    Xbad = Xo.clone()
    Xbad[7, 5] = float('inf')
    for Xin in (Xo, Xbad):
        Sin = hip.spmm(A_own, Xin)
        K1, (s1, b1) = hip.rhs_rk(A1, Xin, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3, X_halo=Xh, **kw)
        K2, (s2, b2) = hip.rhs_rk(A2, Sin, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3, X_halo=Xh, y1=Xin, **kw)
        assert b1 == b2 == (0.0 if Xin is Xo else 1.0)
        if Xin is Xo:
            assert torch.equal(K1, K2) and abs(s1 - s2) <= 1e-12 * abs(s1)
    dt = np.float32(0.37)
    for st in range(4):
        K1, y1_ = hip.rhs_rk(A1, Xo, W, b, 'rk4', y0, ks[:st], [dt], X_halo=Xh, **kw)
        K2, y2_ = hip.rhs_rk(A2, S, W, b, 'rk4', y0, ks[:st], [dt], X_halo=Xh, **kw)
        assert torch.equal(K1, K2) and torch.equal(y1_, y2_)


@pytest.mark.parametrize('no_control', [False, True])
@pytest.mark.parametrize('H', [256, 64])
def test_error_record_split_over_row_blocks_accumulates(dev, H, no_control):
    """A shard's error launch split into row blocks (interior during the exchange, boundary bands after it): every launch
    takes its rows of y1 explicitly and ADDS its record to the device buffer (NDCN_F_ACCUM); one read-back at the end."""
    from ndcn_amd import hip, CsrOperator, graphs
    side = 40
    n = side * side
    m = graphs.normalized_laplacian(graphs.grid_8_neighbor(side)).tocsr()
    g = torch.Generator().manual_seed(5)
    X = torch.rand(n, H, generator=g).to(dev)
    X[900, 3] = float('nan')
    X[30, 1] = float('inf')
    y0 = torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    for Xin, want_bad in ((torch.rand(n, H, generator=g).to(dev), 0.0), (X, 2.0)):
        K, (s, bad) = hip.rhs_rk(CsrOperator.from_scipy(m, dev), Xin, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3,
                                 no_control=no_control)
        Ks = torch.empty_like(K)
        cuts = [(0, 2 * side), (2 * side, n - 3 * side), (n - 3 * side, n)]
        rec = None
        for i, (a, e) in enumerate(cuts):
            op = CsrOperator.from_scipy(m[a:e], dev)
            op.lattice_hint = (a, n)
            _, rec = hip.rhs_rk(op, Xin, W, b, 'error', y0[a:e], [k[a:e] for k in ks], cs, rtol=1e-2, atol=1e-3,
                                no_control=no_control, out_K=Ks[a:e], y1=Xin[a:e], accum=i > 0, fetch=i == len(cuts) - 1)
            assert (rec is None) == (i < len(cuts) - 1)
        assert bad == rec[1] == want_bad
        if want_bad == 0.0:
            # (fused epilogues: fp64 partial sums, any split gives the same bits to 1e-12; the composed path's stand-alone
            # error kernel sums each block in ATen's float32 order: the split then differs by float32 rounding)
            assert torch.equal(K, Ks) and abs(rec[0] - s) <= (1e-12 if H == 256 else 1e-5) * abs(s)
        else:
            assert np.isnan(rec[0]) and np.isnan(s)


@pytest.mark.parametrize('kernel', ['fused3', 'fused2', 'rec', 'wide', 'composed'])
def test_partial_error_sum_in_the_stage6_launch(dev, kernel):
    """dopri5's error launch used to read {y0, k1, k3, k4, k5, k6, y1}; the launch that produces k6 now also writes
    E = sum_{j<=6} dt c_err[j] k_j (y_aux of ndcn_rhs_rk_f32) and the error launch runs with ONE earlier 'stage' E with
    coefficient 1.  Same terms, same order: E equals the separately rounded left-to-right sum bit for bit, and the error
    record of {E, k7} equals the record of the seven-term launch to the last bit of its fp64 sum."""
    from ndcn_amd import hip, CsrOperator, graphs
    H = 64 if kernel == 'composed' else 256
    side = 40
    n = side * side
    m = graphs.normalized_laplacian(graphs.grid_8_neighbor(side)).tocsr()
    A = _no_plan(CsrOperator.from_scipy(m, dev))
    if kernel in ('fused3', 'rec'):
        A.group_order = torch.as_tensor(A.detect_stencil_order(), dtype=torch.int32).to(dev)
        A.build_rec_plan(16, 40, 2)
    no_control = kernel in ('rec', 'wide')
    g = torch.Generator().manual_seed(3)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    ce = [np.float32(c) for c in (0.013, 0.021, -0.017, 0.009, -0.004, 0.025)]
    kw = dict(no_control=no_control)
    for npv in (4, 0, 5):
        c, c2 = cs[:npv] + [cs[5]], ce[:npv] + [ce[4]]
        K, yn, E = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], c, aux_cs=c2, **kw)
        K0, yn0 = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], c, **kw)
        assert torch.equal(K, K0) and torch.equal(yn, yn0)
        zero = torch.zeros_like(y0)
        assert torch.equal(E, hip.combine(zero, ks[:npv] + [K0], c2))              # 0 + s == s exactly
    # the error launch on {E, K7}: one earlier stage with coefficient 1
    K6, y1, E = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:4], cs[:4] + [cs[5]], aux_cs=ce[:4] + [ce[4]], **kw)
    K7a, (sa, ba) = hip.rhs_rk(A, y1, W, b, 'error', y0, [E], [np.float32(1.0), ce[5]], rtol=1e-2, atol=1e-3, **kw)
    K7b, (sb, bb) = hip.rhs_rk(A, y1, W, b, 'error', y0, ks[:4] + [K6], ce[:4] + [ce[4], ce[5]], rtol=1e-2, atol=1e-3, **kw)
    # (the record is a double-precision sum of per-wave partials: identical terms, and identical bits as long as both launches
    # run the same wave split - rhs_fused3.hip: f3_producers)
    assert torch.equal(K7a, K7b) and abs(sa - sb) <= 1e-12 * abs(sb) and ba == bb == 0.0
    # the stage-6 hand-over (solver.hip: enqueue_attempt): the launch that produces k4 also writes
    # P = c1 k1 + c2 k2 + c3 k3 + c4 k4 of the NEXT stage's sum; the launch that produces k5 then forms y0 + (1 * P + c5 k5)
    # from {y0, P} - bit for bit what it forms from {y0, k1, k2, k3, k4}
    b5 = [np.float32(c) for c in (2.846275, -10.757576, 8.906423, 0.278409, -0.273531)]
    K4, y5, P = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:3], cs[:3] + [cs[5]], aux_cs=b5[:4], **kw)
    K5a, y6a = hip.rhs_rk(A, y5, W, b, 'combine', y0, [P], [np.float32(1.0), b5[4]], **kw)
    K5b, y6b = hip.rhs_rk(A, y5, W, b, 'combine', y0, ks[:3] + [K4], b5, **kw)
    assert torch.equal(K5a, K5b) and torch.equal(y6a, y6b)


@pytest.mark.parametrize('side', [40, 47, 64])
def test_first_stage_input_formed_on_the_staged_rows(dev, side):
    """ndcn_rhs_rk_xadd_f32 (ABI 10): the launch that opens a dopri5 step evaluates f(X + c Xadd) with the sum formed on the
    neighbour rows it stages - bit for bit combine(X, [Xadd], [c]) followed by the one-stage COMBINE launch (K and y_next),
    also where groups are not staged (a plan built without the lattice hint is refused: no such kernel), repeated
    (a premature read of an Xadd row would be a race); refused with a halo panel / other modes."""
    from ndcn_amd import hip, CsrOperator, graphs, _lib
    n = side * side
    m = graphs.normalized_laplacian(graphs.grid_8_neighbor(side)).tocsr()
    A = CsrOperator.from_scipy(m, dev)
    g = torch.Generator().manual_seed(side)
    X, y0 = torch.rand(n, 256, generator=g).to(dev), torch.rand(n, 256, generator=g).to(dev)
    k1 = torch.randn(n, 256, generator=g).to(dev)
    W = ((torch.rand(256, 256, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(256, generator=g) - 0.5) / 8).to(dev)
    c, cs = np.float32(0.0371), [np.float32(0.11), np.float32(-0.23)]
    tmp = hip.combine(X, [k1], [c])
    K0, y0n = hip.rhs_rk(A, tmp, W, b, 'combine', y0, [k1], cs)
    assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
    for _ in range(3):
        got = hip.rhs_rk_xadd(A, X, k1, c, W, b, y0, k1, cs)
        assert got is not None
        assert torch.equal(got[0], K0) and torch.equal(got[1], y0n)
    # a different added panel than the earlier stage
    z = torch.randn(n, 256, generator=g).to(dev)
    K1, y1n = hip.rhs_rk(A, hip.combine(X, [z], [c]), W, b, 'combine', y0, [k1], cs)
    got = hip.rhs_rk_xadd(A, X, z, c, W, b, y0, k1, cs)
    assert torch.equal(got[0], K1) and torch.equal(got[1], y1n)
    # groups the record cannot hold (rows with far-away entries): gathered directly inside the kernel, same bits
    rs = np.random.RandomState(side)
    hot = np.repeat(rs.choice(n, 6, replace=False), 12)                     # six rows with 12 far-away entries each
    extra = sp.csr_matrix((rs.rand(72).astype(np.float32), (hot, rs.randint(0, n, 72))), shape=(n, n))
    m2 = (m + extra).tocsr()
    m2.sort_indices()
    A2 = _no_plan(CsrOperator.from_scipy(m2, dev))
    A2.group_order = torch.as_tensor(A.detect_stencil_order() if A.group_order is None else A.group_order.cpu().numpy(), dtype=torch.int32).to(dev)
    staged, _ = A2.build_rec_plan(16, 40, 2)
    assert 0.5 < staged < 1.0
    K2, y2n = hip.rhs_rk(A2, tmp, W, b, 'combine', y0, [k1], cs)
    assert _lib.load().ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
    got = hip.rhs_rk_xadd(A2, X, k1, c, W, b, y0, k1, cs)
    assert got is not None and torch.equal(got[0], K2) and torch.equal(got[1], y2n)
    # no lattice plan: no such kernel
    P = _no_plan(CsrOperator.from_scipy(m, dev))
    assert hip.rhs_rk_xadd(P, X, k1, c, W, b, y0, k1, cs) is None
    lib = _lib.load()
    assert int(lib.ndcn_rhs_xadd_supported(A.view_ref(), 256, _lib.F_RELU, _lib.RK_COMBINE, 2)) == 0
    assert int(lib.ndcn_rhs_xadd_supported(A.view_ref(), 256, _lib.F_RELU, _lib.RK_ERROR, 1)) == 0


@pytest.mark.parametrize('n,H', [(20000, 20), (20000, 64), (6000, 128)])
def test_narrow_panels_beyond_the_one_launch_range_take_the_composed_path(dev, n, H):
    """rhs_small.hip serves n H <= 2^18 (launch-bound sizes); larger narrow panels run row SpMM -> scratch -> MFMA Linear
    (-> stage kernel), whose scratch ndcn_rhs_work_bytes must size from the same decision (it returned the one-launch
    token for every H <= 128: with the one-launch kernel switched off the composed path wrote past it).  Values against
    fp64; the switch NDCN_RHS_SMALL=0 on a small panel in a fresh process."""
    import subprocess, sys
    from ndcn_amd import hip, CsrOperator, _lib
    assert int(_lib.load().ndcn_rhs_work_bytes(n, H, _lib.F_RELU)) == n * H * 4
    assert int(_lib.load().ndcn_rhs_work_bytes(400, H, _lib.F_RELU)) == 16
    m = rand_csr(n, n, 6, seed=H)
    A = CsrOperator.from_scipy(m, dev)
    g = torch.Generator().manual_seed(H)
    X, y0 = torch.randn(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    k1 = torch.randn(n, H, generator=g).to(dev)
    W = ((torch.rand(H, H, generator=g) - 0.5)).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5)).to(dev)
    ref = np.maximum((m.astype(np.float64) @ X.double().cpu().numpy()) @ W.double().cpu().numpy().T + b.double().cpu().numpy(), 0.0)
    K = hip.rhs(A, X, W, b)
    assert float(np.abs(K.double().cpu().numpy() - ref).max()) < 1e-4
    K2, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, [k1], [np.float32(0.3), np.float32(-0.2)])
    assert torch.equal(K2, K)
    assert torch.equal(yn, y0 + (k1 * np.float32(0.3) + K * np.float32(-0.2)))
    code = ("import torch, numpy as np, scipy.sparse as sp\n"
            "from ndcn_amd import hip, CsrOperator\n"
            "dev = torch.device('cuda:0')\n"
            "m = sp.random(400, 400, density=0.02, random_state=np.random.RandomState(0), format='csr', dtype=np.float32)\n"
            "A = CsrOperator.from_scipy(m, dev)\n"
            "X = torch.rand(400, %d, device=dev); W = torch.rand(%d, %d, device=dev) - .5; b = torch.rand(%d, device=dev)\n"
            "K = hip.rhs(A, X, W, b); torch.cuda.synchronize()\n"
            "ref = torch.relu((torch.from_numpy(m.toarray()).to(dev).double() @ X.double()) @ W.double().t() + b.double())\n"
            "assert float((K.double() - ref).abs().max()) < 1e-4\n"
            "print('ok')\n" % (H, H, H, H))
    env = dict(os.environ, NDCN_RHS_SMALL='0', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('H', [1, 16, 20, 33, 64, 65, 100, 128])
def test_narrow_panel_rhs_is_one_launch_and_equals_the_composed_kernels(dev, H):
    """H <= 128 (the README dynamics commands run H = 20): the whole ODEFunc - and the RK algebra consuming it - in ONE
    launch of rhs_small.hip.  Same fma sequences as ndcn_spmm_f32 -> ndcn_linear_f32 -> the stage kernels: bit-identical
    K, y_next, E; error sums to fp64 rounding; halo panel, ragged rows, a 150-entry row, empty rows."""
    from ndcn_amd import hip, CsrOperator, _lib
    n = 1500
    m = rand_csr(n, n, 7, seed=H)
    m = sp.vstack([m[:-1], sp.csr_matrix(np.ones((1, n), dtype=np.float32))[:, :]]).tocsr() if False else m
    long_row = sp.random(1, n, density=0.1, random_state=np.random.RandomState(1), format='csr', dtype=np.float32)
    m = sp.vstack([m[:700], long_row, m[701:]]).tocsr()
    m.sort_indices()
    A = CsrOperator.from_scipy(m, dev)
    g = torch.Generator().manual_seed(H)
    X, y0 = torch.randn(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W = ((torch.rand(H, H, generator=g) - 0.5)).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5)).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    ce = [np.float32(c) for c in (0.013, 0.021, -0.017, 0.009, -0.004, 0.025)]
    lib = _lib.load()
    nk = lib.ndcn_prof_kinds()
    buf = (_lib.ctypes.c_double * (4 * nk))()
    lib.ndcn_prof_enable(1)
    lib.ndcn_prof_read(buf, nk)
    K = hip.rhs(A, X, W, b)
    torch.cuda.synchronize()
    lib.ndcn_prof_enable(0)
    lib.ndcn_prof_read(buf, nk)
    launches = {name: int(buf[4 * i]) for i, name in enumerate(_lib.PROF_KINDS) if buf[4 * i]}
    assert launches == {'rhs_fused': 1}, launches                          # one launch, none of spmm / linear
    K_ref = hip.linear(hip.spmm(A, X), W, b, relu=True)
    assert torch.equal(K, K_ref)
    # halo split
    got = hip.rhs(A, X[:900].contiguous(), W, b, X_halo=X[900:].contiguous())
    assert torch.equal(got, K_ref)
    for npv in (0, 1, 4, 5):
        c = cs[:npv] + [cs[5]]
        K1, yn, E = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:npv], c, aux_cs=ce[:npv] + [ce[5]])
        assert torch.equal(K1, K_ref) and torch.equal(yn, hip.combine(y0, ks[:npv] + [K_ref], c))
        assert torch.equal(E, hip.combine(torch.zeros_like(y0), ks[:npv] + [K_ref], ce[:npv] + [ce[5]]))
        K2, (s1, b1) = hip.rhs_rk(A, X, W, b, 'error', y0, ks[:npv], c, rtol=1e-2, atol=1e-3)
        s2, b2 = hip.error(y0, X, ks[:npv] + [K_ref], c, 1e-2, 1e-3)
        # (the stand-alone kernel sums a panel of this size in ATen's float32 order, the epilogue in fp64)
        assert torch.equal(K2, K_ref) and abs(s1 - s2) <= 1e-5 * abs(s2) and b1 == b2 == 0.0
    dt = np.float32(0.37)
    for st in range(4):
        K3, yn = hip.rhs_rk(A, X, W, b, 'rk4', y0, ks[:st], [dt])
        assert torch.equal(K3, K_ref) and torch.equal(yn, hip.fixed_stage(2 + st, y0, *(ks[:st] + [K_ref]), dt=dt))


def test_halo_exchange_through_the_c_abi_only(dev):
    """ndcn_comm_* / ndcn_halo_plan_* / ndcn_halo_exchange_f32 driven through ctypes alone (no torch.distributed): a
    one-rank RCCL communicator whose plan routes rows of the own panel through the exchange to itself - the grouped
    ncclSend / ncclRecv really executes on the 1-GPU box - plus the controller's all-reduce (identity at world 1)."""
    import ctypes
    from ndcn_amd import _lib
    lib = _lib.load()
    idbuf = ctypes.create_string_buffer(128)
    _lib.check(lib.ndcn_comm_unique_id(idbuf))
    comm = ctypes.c_void_p()
    _lib.check(lib.ndcn_comm_create(idbuf, 1, 0, ctypes.byref(comm)))
    try:
        n, H, k = 5000, 256, 777
        X = torch.randn(n, H, device=dev)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:k].to(torch.int32).to(dev)
        L1 = ctypes.c_int64 * 1
        plan = ctypes.c_void_p()
        _lib.check(lib.ndcn_halo_plan_create(comm, k, L1(k), L1(k), _lib.ptr(idx), 1, ctypes.byref(plan)))
        pack = torch.empty(k, H, device=dev)
        halo = torch.full((k, H), float('nan'), device=dev)
        _lib.check(lib.ndcn_halo_exchange_f32(plan, _lib.ptr(X), H, _lib.ptr(pack), _lib.ptr(halo), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(halo, X[idx.long()])
        # a plan nobody moves rows in skips the collective on every rank
        plan0 = ctypes.c_void_p()
        _lib.check(lib.ndcn_halo_plan_create(comm, 0, L1(0), L1(0), None, 0, ctypes.byref(plan0)))
        _lib.check(lib.ndcn_halo_exchange_f32(plan0, _lib.ptr(X), H, None, None, _lib.stream_ptr()))
        # receive counts must add up to n_halo
        bad = ctypes.c_void_p()
        assert lib.ndcn_halo_plan_create(comm, k + 1, L1(k), L1(k), _lib.ptr(idx), 1, ctypes.byref(bad)) == _lib.EINVAL
        rec = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
        _lib.check(lib.ndcn_comm_allreduce_sum_f64(comm, _lib.ptr(rec), 2, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert rec.tolist() == [1.5, 2.0]
        lib.ndcn_halo_plan_destroy(plan)
        lib.ndcn_halo_plan_destroy(plan0)
    finally:
        lib.ndcn_comm_destroy(comm)


def test_long_row_plan_equals_in_kernel_gather(dev):
    """Power-law graph: rows longer than the plan's threshold are evaluated by the segment SpMMs ahead of the fused
    kernel and enter it as one entry of a second panel - same results as gathering them inside the kernel."""
    from ndcn_amd import hip, CsrOperator, graphs
    H, n = 256, 6000
    m = graphs.normalized_laplacian(graphs.make_graph('power_law', n, seed=1)).tocsr()
    deg = np.diff(m.indptr)
    assert deg.max() > 300                                     # hubs spanning several 256-entry segments
    g = torch.Generator().manual_seed(0)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    A_plan = CsrOperator.from_scipy(m, dev)
    A_plan.ensure_plans(H)
    assert A_plan.hub is not None and A_plan.hub['n'] == int((deg > A_plan.hub['threshold']).sum()) > 0
    assert A_plan.hub['nseg'] > A_plan.hub['n']                # at least one hub spans several segments
    A_ref = CsrOperator.from_scipy(m, dev)
    _no_plan(A_ref)                                            # no plans: every row is gathered inside the kernel
    ref = hip.rhs(A_ref, X, W, b)
    assert torch.allclose(hip.rhs(A_plan, X, W, b), ref, rtol=1e-5, atol=1e-6)
    exact = torch.relu(torch.from_numpy((m.astype(np.float64) @ X.cpu().double().numpy())).to(dev) @ W.double().T + b.double())
    assert (ref.double() - exact).abs().max() < 2e-5 and (hip.rhs(A_plan, X, W, b).double() - exact).abs().max() < 2e-5
    K1, y1 = hip.rhs_rk(A_plan, X, W, b, 'combine', y0, ks[:4], cs[:4] + [cs[5]])
    K2, y2 = hip.rhs_rk(A_ref, X, W, b, 'combine', y0, ks[:4], cs[:4] + [cs[5]])
    assert torch.allclose(K1, K2, rtol=1e-5, atol=1e-6) and torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
    _, (s1, b1) = hip.rhs_rk(A_plan, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
    _, (s2, b2) = hip.rhs_rk(A_ref, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
    assert abs(float(s1) - float(s2)) <= 1e-6 * abs(float(s2)) and float(b1) == float(b2) == 0.0


@pytest.mark.parametrize('plans', [False, True])
def test_fused_rhs_every_row_length(dev, plans):
    """Rows of 0, 1, 2, ... 70 entries (every split of the one-round gather: pieces 8/4/2/1, rounds of 16, rows past
    the long-row threshold), ragged last tile, with and without the operator's plans - against fp64."""
    from ndcn_amd import hip, CsrOperator
    H, n = 256, 71 * 13 + 5
    rng = np.random.RandomState(7)
    deg = np.arange(n) % 71
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([rng.choice(n, size=d, replace=False) for d in deg])
    m = sp.csr_matrix((rng.randn(rows.size).astype(np.float32) / 8, (rows, cols)), shape=(n, n))
    m.sort_indices()
    A = CsrOperator.from_scipy(m, dev)
    if plans:
        os.environ['NDCN_HUB_THRESHOLD'] = '32'
        try:
            A.ensure_plans(H)
        finally:
            del os.environ['NDCN_HUB_THRESHOLD']
        assert A.hub is not None
    else:
        _no_plan(A)
    g = torch.Generator().manual_seed(3)
    X = torch.rand(n, H, generator=g).to(dev)
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    S = torch.from_numpy(m.astype(np.float64) @ X.cpu().double().numpy()).to(dev)
    exact = torch.relu(S @ W.double().T + b.double())
    got = hip.rhs(A, X, W, b)
    assert (got.double() - exact).abs().max() < 2e-5
    y0 = torch.rand(n, H, generator=g).to(dev)
    K, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, [X], [np.float32(0.25), np.float32(-0.5)])
    assert torch.equal(K, got)
    assert torch.equal(yn, y0 + (X * 0.25 + K * -0.5))


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 127, 129, 2049])
def test_fused_rhs_tiny_and_ragged_sizes(dev, n):
    """Fewer rows than one tile, exactly one tile, one row into the next tile, fewer tiles than workgroups: plain,
    COMBINE and ERROR launches against fp64 / the separate kernels."""
    from ndcn_amd import hip, CsrOperator
    H = 256
    m = rand_csr(n, n, 6, seed=n)
    A = CsrOperator.from_scipy(m, dev)
    g = torch.Generator().manual_seed(n)
    X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
    S = torch.from_numpy(m.astype(np.float64) @ X.cpu().double().numpy()).to(dev)
    exact = torch.relu(S @ W.double().T + b.double())
    K = hip.rhs(A, X, W, b)
    assert (K.double() - exact).abs().max() < 2e-5
    K2, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, ks[:2], cs[:2] + [cs[5]])
    assert torch.equal(K2, K) and torch.equal(yn, hip.combine(y0, ks[:2] + [K], cs[:2] + [cs[5]]))
    K3, (ss, bad) = hip.rhs_rk(A, X, W, b, 'error', y0, ks, cs, rtol=1e-2, atol=1e-3)
    s_ref, bad_ref = hip.error(y0, X, ks + [K], cs, 1e-2, 1e-3)
    assert torch.equal(K3, K) and float(bad) == float(bad_ref) == 0.0
    assert abs(float(ss) - float(s_ref)) <= 1e-5 * max(abs(float(s_ref)), 1e-30)    # (stand-alone: ATen float32 order below 2^20 elements)


def test_fused_rk4_stage_epilogues_bitwise_vs_separate_kernels(dev):
    """ndcn_rhs_rk_f32 in NDCN_RK_RK4 mode (stage algebra of rk4_alt_step_func in the RHS epilogue) against
    ODEFunc + ndcn_fixed_stage_f32 ops 2-5 (which are pinned to the reference's operator order)."""
    from ndcn_amd import hip, CsrOperator, graphs
    H, side = 256, 45
    n = side * side
    A = CsrOperator.from_scipy(graphs.normalized_laplacian(graphs.grid_8_neighbor(side)), dev)
    g = torch.Generator().manual_seed(2)
    X, y = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
    ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(3)]
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    dt = np.float32(0.0505)
    K = hip.rhs(A, X, W, b)
    for i in range(4):
        Ki, out = hip.rhs_rk(A, X, W, b, 'rk4', y, ks[:i], [dt])
        assert torch.equal(Ki, K)
        args = (ks[:i] + [K] + [None] * 3)[:4]
        want = hip.fixed_stage(2 + i, y, args[0], args[1], args[2], args[3], dt=dt)
        assert torch.equal(out, want), i


def test_fused_rhs_panel_beyond_2_gib(dev):
    """2.2 M nodes x 256 floats = 2.25 GB per panel: row offsets of the fused kernel's buffer accesses pass 2^31
    (they are unsigned 32-bit: the kernel serves panels < 4 GiB).  Checked on row samples against fp64."""
    from ndcn_amd import hip, CsrOperator, graphs
    H, R, C = 256, 2200, 1000
    m = graphs.grid_operator_row_block(R, C, 0, R, 'norm_lap').tocsr()
    n = R * C
    assert n * H * 4 > 2 ** 31
    A = CsrOperator.from_scipy(m, dev)
    g = torch.Generator().manual_seed(0)
    W, b = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev), ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    X = torch.rand(n, H, device=dev)
    y0 = torch.rand(n, H, device=dev)
    k1 = torch.rand(n, H, device=dev)
    c = [np.float32(0.3), np.float32(-0.2)]
    K, yn = hip.rhs_rk(A, X, W, b, 'combine', y0, [k1], c)
    rows = torch.tensor([0, 1, 999, 1000, 1048575, 2097151, 2097152, 2150000, n - 1001, n - 1], device=dev)
    sub = m[rows.cpu().numpy()].tocsr()
    cols = np.unique(sub.indices)
    small = sp.csr_matrix((sub.data.astype(np.float64), np.searchsorted(cols, sub.indices), sub.indptr), shape=(sub.shape[0], cols.size))
    S = torch.from_numpy(small @ X[torch.from_numpy(cols).to(dev)].cpu().double().numpy()).to(dev)
    exact = torch.relu(S @ W.double().T + b.double())
    assert (K[rows].double() - exact).abs().max() < 2e-5
    want = y0[rows] + (k1[rows] * float(c[0]) + K[rows] * float(c[1]))
    assert torch.equal(yn[rows], want)


def test_first_generation_fused_kernel_still_serves_as_fallback(dev):
    """NDCN_RHS_FUSED2=0 (what panels >= 4 GiB fall back to): rhs golden + a dopri5 golden in a fresh process."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_golden\n"
        "from ndcn_amd import hip, CsrOperator\n"
        "d = load_golden('rhs_grid400_H256_default_coo'); dev = torch.device('cuda:0')\n"
        "A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)\n"
        "T = lambda a: torch.from_numpy(np.asarray(a)).to(dev)\n"
        "out = hip.rhs(A, T(d['x']), T(d['W']), T(d['b'])).cpu().numpy()\n"
        "assert np.abs(out - d['out']).max() <= 2e-5\n"
        "K, yn = hip.rhs_rk(A, T(d['x']), T(d['W']), T(d['b']), 'combine', T(d['x']), [], [np.float32(0.5)])\n"
        "assert np.abs(K.cpu().numpy() - d['out']).max() <= 2e-5\n"
        "assert torch.equal(yn, T(d['x']) + K * 0.5)\n"
        "print('fallback ok')\n") % (ROOT, os.path.join(ROOT, 'tests'))
    env = dict(os.environ, NDCN_RHS_FUSED2='0')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'fallback ok' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('Hi,Ho', [(20, 20), (256, 256), (33, 7)])
def test_gcn_entry_point_is_linear_then_spmm(dev, Hi, Ho):
    """ndcn_gcn_f32 = GraphConvolution.forward of the reference (models.py:14-18: fc, then torch.sparse.mm) in one call: the same
    bits as ndcn_linear_f32 followed by ndcn_spmm_f32, and the oracle's expression to rounding."""
    from ndcn_amd import graphs, hip
    L = graphs.normalized_adj(graphs.make_graph('power_law', 1500, seed=4))
    A = graphs.to_device(L, dev)
    gen = torch.Generator().manual_seed(Hi)
    X, W, b = torch.randn(1500, Hi, generator=gen), torch.randn(Ho, Hi, generator=gen) / 4, torch.randn(Ho, generator=gen)
    got = hip.gcn(A, X.to(dev), W.to(dev), b.to(dev))
    assert torch.equal(got, hip.spmm(A, hip.linear(X.to(dev), W.to(dev), b.to(dev))))
    ref = orc.gcn_layer(orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape), X, W, b)
    assert float((got.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_integration_stub_runs(dev):
    """the ~40-line ctypes stub a reference maintainer would add (examples/ndcn_hip_binding.py,
    no ndcn_amd import: ndcn_csr_create on the arrays of a torch COO operator, then ndcn_rhs_f32 on the handle's view)
    against the reference's own output on the 20 x 20 grid - and it must reach rhs_fused3 at H = 256."""
    import importlib.util
    from ndcn_amd import _lib
    os.environ['NDCN_HIP_LIB'] = _lib.LIB_PATH
    spec = importlib.util.spec_from_file_location('ndcn_hip_binding', os.path.join(ROOT, 'examples', 'ndcn_hip_binding.py'))
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    d = load_golden('rhs_grid400_H256_default_coo')
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape']).to(dev)
    op = stub.Operator(A, 256)
    out = stub.odefunc_forward(op, T(d['x']).to(dev), T(d['W']).to(dev), T(d['b']).to(dev))
    assert np.abs(out.cpu().numpy() - d['out']).max() <= 2e-5
    # the handle built the lattice's group-record plan inside the library: the binding lands on the kernel the bench times
    assert stub.last_rhs_path() == 2                                       # NDCN_PATH_FUSED3
    d = load_golden('rhs_grid400_H20_no_control_coo')
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape']).to(dev)
    op = stub.Operator(A, 20)
    out = stub.odefunc_forward(op, T(d['x']).to(dev), T(d['W']).to(dev), T(d['b']).to(dev), no_control=True)
    assert np.abs(out.cpu().numpy() - d['out']).max() <= 2e-5


# ------------------------------------------------------------------------------------------- RK bookkeeping
@pytest.mark.parametrize('shape', [(400, 20), (1001, 1), (257, 3), (4096, 256)])
def test_rk_kernels_bitwise_vs_reference_op_order(dev, shape):
    """combine / interp / fixed-stage kernels reproduce the reference's separately-rounded op chains."""
    from ndcn_amd import hip
    torch.manual_seed(shape[0])
    y0, y1 = torch.randn(shape), torch.randn(shape)
    ks = [torch.randn(shape) for _ in range(7)]
    cs = [np.float32(c) for c in np.random.RandomState(0).randn(7)]
    g = lambda x: x.to(dev)
    for n in (1, 2, 5, 7):
        got = hip.combine(g(y0), [g(k) for k in ks[:n]], cs[:n]).cpu()
        assert torch.equal(got, OracleOps.combine(y0, ks[:n], cs[:n]))
    dt = np.float32(0.37)
    cmid = [np.float32(dt * np.float32(c)) for c in orc.DP_C_MID]
    got = hip.interp_fit(g(y0), g(y1), [g(k) for k in ks], cmid, dt)
    ref = OracleOps.interp_fit(y0, y1, [k for k, c in zip(ks, cmid)], cmid, dt)
    for a, b in zip(got, ref):
        assert torch.equal(a.cpu(), b)
    x = np.float32(0.3)
    xp = (np.float32(x * x * x * x), np.float32(x * x * x), np.float32(x * x), x, np.float32(1))
    got = hip.interp_eval(tuple(g(v) for v in ref), g(y0), xp).cpu()
    assert torch.equal(got, OracleOps.interp_eval(ref, y0, xp))
    # single-pass variant (what the device solver uses for the first tick inside a step)
    direct = hip.interp_direct(g(y0), g(y1), [g(k) for k in ks], cmid, dt, xp).cpu()
    assert torch.equal(direct, got)
    for op in range(6):
        got = hip.fixed_stage(op, g(y0), g(ks[0]), g(ks[1]), g(ks[2]), g(ks[3]), dt=dt).cpu()
        assert torch.equal(got, OracleOps.fixed_stage(op, y0, ks[0], ks[1], ks[2], ks[3], dt=dt)), op


@pytest.mark.parametrize('n', [1, 63, 4096, 262144, 1000003])
def test_reductions(dev, n):
    from ndcn_amd import hip
    torch.manual_seed(n)
    y0, y1, a, b = (torch.randn(n) for _ in range(4))
    ks = [torch.randn(n) for _ in range(6)]
    cs = [np.float32(c) for c in (0.1, -0.2, 0.3, 0.05, -0.07, 0.01)]
    g = lambda x: x.to(dev)
    s, bad = hip.error(g(y0), g(y1), [g(k) for k in ks], cs, 1e-2, 1e-3)
    rs, rbad = OracleOps.error(y0, y1, ks, cs, np.float32(1e-2), np.float32(1e-3))
    # 8 <= n <= 2^18 (rk.hip: aten_order_max_elems - the reference-sized panels): the float32 sum torch.mean forms (ATen's
    # cascade order, torch 2.10 AVX2 kernels), bit for bit; outside that range fp64 partial sums in a fixed order
    exact = 8 <= n <= (1 << 18)
    assert bad == 0 and (s == rs if exact else abs(s - rs) <= (1e-9 if n < 8 else 1e-6) * abs(rs))
    # the float32 NORM (the square root of the sum) must equal torch's bit for bit in that range: the kernel adds up in
    # ATen's order (8 fma chains, then the tail)
    nrm = lambda v: np.float32(np.sqrt(v))
    # (beyond the range the kernel sums in fp64; torch's own float32 chains are then the less accurate side: ~1e-5 at 10^6 terms)
    same = (lambda u, v: u == v) if n <= (1 << 18) else (lambda u, v: abs(float(u) - float(v)) <= 1e-4 * abs(float(v)))
    s, bad = hip.scaled_sumsq(g(a), g(b), g(y0), 1e-2, 1e-3)
    q = (a - b) / (np.float32(1e-3) + torch.abs(y0) * np.float32(1e-2))
    assert same(nrm(s), np.float32(q.norm().item()))
    s, bad = hip.scaled_sumsq(g(a), None, g(y0), 1e-2, 1e-3)
    q = a / (np.float32(1e-3) + torch.abs(y0) * np.float32(1e-2))
    assert same(nrm(s), np.float32(q.norm().item()))
    # determinism: same bits on a second run
    assert hip.scaled_sumsq(g(a), None, g(y0), 1e-2, 1e-3)[0] == s
    # non-finite detection
    y1[n // 2] = float('inf')
    a[0] = float('nan')
    assert hip.error(g(y0), g(y1), [g(k) for k in ks], cs, 1e-2, 1e-3)[1] == 1
    assert hip.scaled_sumsq(g(a), None, g(y0), 1e-2, 1e-3)[1] == 1


# ------------------------------------------------------------------------------------------- truth dynamics
def test_truth_rhs_kernels(dev):
    from ndcn_amd import hip, CsrOperator
    d = load_golden('truth_mutual_coo')
    n = int(d['n'])
    A = CsrOperator.from_arrays(d['A_indptr'], d['A_indices'], d['A_data'], (n, n), dev)
    Ac = orc.coo_from_csr(d['A_indptr'], d['A_indices'], d['A_data'], (n, n))
    for k in (0, 3, 20):
        x = T(d['traj'][k])
        got = hip.mutual_rhs(A, x.to(dev)).cpu()
        ref = orc.mutual_rhs(Ac, x)
        assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
        got = hip.gene_rhs(A, x.to(dev)).cpu()
        ref = orc.gene_rhs(Ac, x)
        assert (got - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_packed_weight_cache_follows_weight_updates(dev):
    """ops keeps the packed (split-fp16) image of W between no-grad calls (NDCN_F_PACKED); an in-place update of W must
    be seen by the next call."""
    from ndcn_amd import hip, CsrOperator, graphs
    H = 256
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(40))
    A = CsrOperator.from_scipy(L, dev)
    g = torch.Generator().manual_seed(8)
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = torch.zeros(H, device=dev)
    X = torch.rand(1600, H, generator=g).to(dev)
    with torch.no_grad():
        k1 = hip.rhs(A, X, W, b)
        assert torch.equal(hip.rhs(A, X, W, b), k1)                 # second call: cached image
        W.mul_(-1.5)                                                 # in-place: version counter moves
        k2 = hip.rhs(A, X, W, b)
        ref = torch.relu(hip.spmm(A, X) @ W.t())
        assert float((k2 - ref).abs().max()) < 1e-4 and not torch.equal(k1, k2)
        K, _ = hip.rhs_rk(A, X, W, b, 'combine', X, [], [np.float32(0.5)])
        assert torch.equal(K, k2)
        # a write that bypasses the version counter is NOT seen (documented) until the cache is invalidated
        W.data.mul_(2.0)
        assert torch.equal(hip.rhs(A, X, W, b), k2)
        from ndcn_amd.ops import invalidate_packed_weights
        invalidate_packed_weights()
        k3 = hip.rhs(A, X, W, b)
        assert float((k3 - torch.relu(hip.spmm(A, X) @ W.t())).abs().max()) < 2e-4 and not torch.equal(k3, k2)


def test_packed_weight_cache_sees_data_writes_between_solves(dev):
    """Round-4 advisor: `p.data.add_()` (hand-written SGD, EMA, clipping) does not move the version counter, and the training path
    hands the live Parameter to the cache.  Every odeint call opens a new epoch of the cache: the solve after such a write - forward
    values AND gradients - equals a fresh module holding the same numbers."""
    from ndcn_amd import CsrOperator, graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    H = 256
    A = CsrOperator.from_scipy(graphs.normalized_laplacian(graphs.grid_8_neighbor(24)), dev)
    torch.manual_seed(3)
    f = ODEFunc(H, A).to(dev)
    x0 = torch.rand(576, H, device=dev)
    t = torch.tensor([0., 0.3, 0.6], device=dev)

    def solve(func, grad):
        if not grad:
            with torch.no_grad():
                return ode.odeint(lambda tt, y: func(tt, y), x0, t, method='rk4'), None     # python-stepped path: hip.rhs per stage
        for p in func.parameters():
            p.grad = None
        y = ode.odeint(func, x0, t, rtol=1e-3, atol=1e-4, method='dopri5')
        y[-1].square().sum().backward()
        return y.detach(), [p.grad.clone() for p in func.parameters()]

    for grad in (False, True):
        solve(f, grad)                                               # packs W (and, with grad, the planes of W^T)
        with torch.no_grad():
            f.wt.weight.data.mul_(0.5)                               # bypasses the version counter
            f.wt.weight.data[7, 9] += 0.25
        got, gg = solve(f, grad)
        fresh = ODEFunc(H, A).to(dev)
        fresh.load_state_dict(f.state_dict())
        want, gw = solve(fresh, grad)
        assert torch.equal(got, want), grad
        if grad:
            assert all(torch.equal(a, b) for a, b in zip(gg, gw))


def test_packed_weight_cache_is_not_fooled_by_address_reuse(dev):
    """A NEW weight tensor that lands on the address of a freed one (same shape: the caching allocator hands the block
    out again) must not hit the packed image of the old one."""
    from ndcn_amd import hip, CsrOperator, graphs
    H = 256
    A = CsrOperator.from_scipy(graphs.normalized_laplacian(graphs.grid_8_neighbor(40)), dev)
    X = torch.rand(1600, H, device=dev)
    b = torch.zeros(H, device=dev)
    seen = set()
    with torch.no_grad():
        for i in range(4):
            W = ((torch.rand(H, H, generator=torch.Generator().manual_seed(i)) - 0.5) / 8).to(dev)
            seen.add(W.data_ptr())
            k = hip.rhs(A, X, W, b)
            assert float((k - torch.relu(hip.spmm(A, X) @ W.t())).abs().max()) < 1e-4, i
            del W
    assert len(seen) < 4                                            # the allocator did reuse an address


@pytest.mark.parametrize('mode,n_prev', [('combine', 1), ('combine', 2), ('combine', 3), ('combine', 4), ('error', 5)])
@pytest.mark.parametrize('side', [24, 112, 400])        # (400: panels beyond 128 MiB - the launches with non-temporal stores)
def test_adjoint_halves_of_the_fused_launch_equal_the_composed_kernels(dev, side, mode, n_prev):
    """ndcn_rhs_rk_adj_f32 (ABI 13): the forward half also writes S = A X - bit-equal to ndcn_spmm_f32, K / y_next / the error record
    unchanged; the transposed half gathers X (.) [M > 0] - bit-equal to the same launch over the panel ndcn_relu_bwd_f32 writes."""
    from ndcn_amd import hip, CsrOperator, graphs, _lib
    H = 256
    A = CsrOperator.from_scipy(graphs.normalized_laplacian(graphs.grid_8_neighbor(side)), dev)
    n = side * side
    g = torch.Generator().manual_seed(side + n_prev)
    X = torch.rand(n, H, generator=g).to(dev)
    M = (torch.rand(n, H, generator=g) - 0.4).to(dev)
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8).to(dev)
    b = ((torch.rand(H, generator=g) - 0.5) / 8).to(dev)
    y0 = torch.rand(n, H, generator=g).to(dev)
    kprev = [torch.rand(n, H, generator=g).to(dev) for _ in range(n_prev)]
    cs = [np.float32(0.1 * (j + 1)) for j in range(n_prev + 1)]
    assert hip.rhs_adj_supported(A, H, mode, n_prev)
    with torch.no_grad():
        tol = (1e-2, 1e-3) if mode == 'error' else (0.0, 0.0)
        ref = hip.rhs_rk(A, X, W, b, mode, y0, kprev, cs, *tol)
        S = torch.empty_like(X)
        got = hip.rhs_rk(A, X, W, b, mode, y0, kprev, cs, *tol, s_out=S)
        assert int(_lib.load().ndcn_debug_last_rhs_path()) == _lib.PATH_FUSED3
        assert torch.equal(got[0], ref[0]) and torch.equal(S, hip.spmm(A, X))
        assert torch.equal(got[1], ref[1]) if mode == 'combine' else got[1] == ref[1]
        Wt = W.t().contiguous()
        kw = dict(relu=False, y1=X) if mode == 'error' else dict(relu=False)
        ref = hip.rhs_rk(A, hip.relu_bwd(X, M), Wt, None, mode, y0, kprev, cs, *tol, **kw)
        got = hip.rhs_rk(A, X, Wt, None, mode, y0, kprev, cs, *tol, x_mask=M, **kw)
        assert torch.equal(got[0], ref[0])
        assert torch.equal(got[1], ref[1]) if mode == 'combine' else got[1] == ref[1]
