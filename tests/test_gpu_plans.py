"""ndcn_csr_create (ndcn_amd/csrc/csr_plan.hip) against the torch / numpy restatement of the plan builders
(tests/_plan_reference.py): the plans are integer work on the CSR arrays, so every array must agree bit for bit and every
decision (lattice or not, which record shape, which hub threshold) must be the same."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from _plan_reference import PlanReference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _pull(ptr, n, dtype, dev):
    """n 4-byte words at a device address of the library's, as a numpy array."""
    from ndcn_amd import _lib
    out = torch.empty(n, dtype=dtype, device=dev)
    if n:
        _lib.check(_lib.load().ndcn_copy_f32(out.data_ptr(), ptr, n, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _graphs():
    from ndcn_amd import graphs
    rng = np.random.RandomState(0)
    grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(37))                 # side not a multiple of the patch
    rect = graphs.normalized_laplacian(graphs.grid_8_neighbor_rect(70, 130))       # strips: PY > 32
    rnd = sp.random(1369, 1369, density=0.01, random_state=rng, format='csr', dtype=np.float32)
    longrow = sp.vstack([grid[:500], sp.csr_matrix(np.ones((1, 1369), np.float32)), grid[501:]]).tocsr()
    mixed = sp.vstack([grid[:800], rnd[800:]]).tocsr()
    five = sp.diags([1., 1., 1., 1., 1.], [-50, -1, 0, 1, 50], shape=(2500, 2500), format='csr', dtype=np.float32)
    out = {'grid37': grid, 'rect70x130': rect, 'random': rnd, 'longrow': longrow, 'mixed': mixed, 'five_point': five,
           'small_world': graphs.normalized_laplacian(graphs.make_graph('small_world', 6000, seed=3)),
           'power_law': graphs.normalized_laplacian(graphs.make_graph('power_law', 8000, seed=2)),
           'gnp': graphs.normalized_laplacian(graphs.make_graph('random', 3000, seed=1)),
           'tiny': graphs.normalized_laplacian(graphs.grid_8_neighbor(5)),
           'empty': sp.csr_matrix((300, 300), dtype=np.float32)}
    for m in out.values():
        m.sort_indices()
    return out


def _compare(A, ref, dev):
    """CsrOperator A (plans by the library) vs PlanReference ref (plans by the restatement)."""
    v = A.view()
    assert (A.rec is None) == (ref.rec is None), (A.rec, None if ref.rec is None else {k: ref.rec[k] for k in ('rows', 'cap', 'staged', 'loads_per_row')})
    if ref.rec is not None:
        for k in ('rows', 'cap', 'kib', 'groups'):
            assert A.rec[k] == ref.rec[k], k
        assert A.rec['staged'] == ref.rec['staged'] and A.rec['loads_per_row'] == ref.rec['loads_per_row']
        words = ref.rec['kib'] * 256
        got = _pull(v.rec, ref.rec['groups'] * words, torch.int32, dev).reshape(-1, words)
        assert np.array_equal(got, ref.rec['rec'].cpu().numpy())
    assert A.stencil_stride == ref.stencil_stride
    if ref.group_order is not None and ref.stencil_stride:
        assert np.array_equal(A.group_order.cpu().numpy(), ref.group_order.cpu().numpy())
    assert bool(v.tile_order) == (ref.tile_order is not None)
    if ref.tile_order is not None:
        assert np.array_equal(_pull(v.tile_order, ref.tile_order.numel(), torch.int32, dev), ref.tile_order.cpu().numpy())
    assert (A.hub is None) == (ref.hub is None)
    if ref.hub is not None:
        h = ref.hub
        for k in ('n', 'nseg', 'H', 'nnz', 'lt_nnz', 'threshold'):
            assert A.hub[k] == h[k], k
        assert (v.hub_n, v.hub_nseg, v.hub_H, v.hub_nnz, v.lt_nnz) == (h['n'], h['nseg'], h['H'], h['nnz'], h['lt_nnz'])
        for name, ptr, n, dt in (('seg_rowptr', v.hub_seg_rowptr, h['nseg'] + 1, torch.int32), ('colidx', v.hub_colidx, h['nnz'], torch.int32),
                                 ('val', v.hub_val, h['nnz'], torch.float32), ('cmb_rowptr', v.hub_cmb_rowptr, h['n'] + 1, torch.int32),
                                 ('cmb_colidx', v.hub_cmb_colidx, h['nseg'], torch.int32), ('cmb_val', v.hub_cmb_val, h['nseg'], torch.float32),
                                 ('lt_rowptr', v.lt_rowptr, A.shape[0] + 1, torch.int32), ('lt_colidx', v.lt_colidx, h['lt_nnz'], torch.int32),
                                 ('lt_val', v.lt_val, h['lt_nnz'], torch.float32)):
            assert np.array_equal(_pull(ptr, n, dt, dev), h[name].cpu().numpy()), name
        n_halo = int(getattr(A, 'n_halo', 0))
        assert A.hub['halo_S'].shape == (n_halo + h['n'], 256) and v.hub_S == A.hub['halo_S'].data_ptr() + 4 * 256 * n_halo
        assert v.hub_Sseg == A.hub['Sseg'].data_ptr()


@pytest.mark.parametrize('name', ['grid37', 'rect70x130', 'random', 'longrow', 'mixed', 'five_point', 'small_world', 'power_law',
                                  'gnp', 'tiny', 'empty'])
def test_automatic_plans_equal_the_restatement(dev, name):
    """ensure_plans(256) = ndcn_csr_create with no hints: lattice detection, walk orders, the shape that pays, the hub
    threshold - the same decisions and the same arrays as the Python builders took."""
    from ndcn_amd import CsrOperator
    m = _graphs()[name]
    A = CsrOperator.from_scipy(m, dev).ensure_plans(256)
    ref = PlanReference.of(CsrOperator.from_scipy(m)).ensure_plans(256)
    _compare(A, ref, dev)
    if name in ('grid37', 'rect70x130', 'five_point'):
        assert A.rec is not None and A.rec['rows'] == 16 and A.stencil_stride in (37, 130, 50)
    if name == 'small_world':
        assert A.rec is not None and (A.rec['rows'], A.rec['cap']) == (8, 48)
    if name == 'power_law':
        assert A.hub is not None and A.rec is None
    if name in ('random', 'gnp', 'empty'):
        assert A.rec is None and A.hub is None


@pytest.mark.parametrize('name', ['grid37', 'random', 'longrow', 'mixed', 'small_world'])
@pytest.mark.parametrize('shape', [(8, 32, 1), (16, 40, 2), (8, 48, 2)])
@pytest.mark.parametrize('hinted', [False, True])
def test_forced_record_shapes_equal_the_restatement(dev, name, shape, hinted):
    """Every record shape on every kind of operator, with and without a lattice walk order: groups that do not fit are
    flagged the same way, column lists, headers and re-indexed entries are identical."""
    from ndcn_amd import CsrOperator, graphs
    gs = _graphs()
    m = gs[name]
    A = CsrOperator.from_scipy(m, dev)
    ref = PlanReference.of(CsrOperator.from_scipy(m))
    if hinted:
        if m.shape[0] != 1369:
            pytest.skip('the lattice order of the 37 x 37 grid')
        order = CsrOperator.from_scipy(gs['grid37'], dev).detect_stencil_order()
        want = PlanReference.of(CsrOperator.from_scipy(gs['grid37'])).detect_stencil_order()
        assert np.array_equal(order, want)
        A.group_order = torch.as_tensor(order, dtype=torch.int32).to(dev)
        ref.group_order = torch.as_tensor(want, dtype=torch.int32)
    got = A.build_rec_plan(*shape)
    exp = ref.build_rec_plan(*shape)
    assert got == exp
    ref.stencil_stride = A.stencil_stride = 0
    ref.tile_order = None
    _compare(A, ref, dev)


def test_row_blocks_of_a_sharded_lattice_equal_the_restatement(dev):
    """A shard's row blocks say where they sit (lattice hint) and how many halo rows precede the hubs' scratch: interior,
    top band (sees only the lattice row BELOW among its own columns), bottom band (only the row above), whole shard."""
    from ndcn_amd import CsrOperator, graphs
    R, C = 40, 80                                              # (bands of >= 64 rows: the detection's minimum)
    full = graphs.normalized_laplacian(graphs.grid_8_neighbor_rect(3 * R, C)).tocsr()
    lo, hi = R * C, 2 * R * C                                  # the middle shard
    blk = full[lo:hi].tocsc()
    own = blk[:, lo:hi]
    halo_cols = np.unique(np.r_[blk[:, :lo].nonzero()[1], hi + blk[:, hi:].nonzero()[1]])
    local = sp.hstack([own, blk[:, halo_cols]]).tocsr()
    local.sort_indices()
    n = hi - lo
    for a, b, with_halo in ((0, C, True), (C, n - C, False), (n - C, n, True), (0, n, True)):
        sub = local[a:b] if with_halo else local[a:b][:, :n]
        sub = sub.tocsr()
        sub.sort_indices()
        A = CsrOperator.from_scipy(sub, dev)
        A.lattice_hint = (a, n)
        A.ensure_plans(256)
        ref = PlanReference.of(CsrOperator.from_scipy(sub), lattice_hint=(a, n)).ensure_plans(256)
        _compare(A, ref, dev)
        assert A.rec is not None and A.rec['rows'] == 16 and A.stencil_stride == C, (a, b)


def test_long_row_plan_behind_a_halo_panel_and_thresholds(dev):
    """n_halo moves the hubs' scratch behind the halo rows; explicit thresholds and NDCN_PLAN_NO_HUB are honoured."""
    from ndcn_amd import CsrOperator, graphs, _lib
    m = graphs.normalized_laplacian(graphs.make_graph('power_law', 8000, seed=2))
    A = CsrOperator.from_scipy(m, dev)
    A.n_halo = 77
    A.ensure_plans(256)
    ref = PlanReference.of(CsrOperator.from_scipy(m), n_halo=77).ensure_plans(256)
    _compare(A, ref, dev)
    assert A.hub['halo_S'].shape[0] == 77 + A.hub['n']
    for thr in (16, 200):
        B = CsrOperator.from_scipy(m, dev).build_plans(256, hub_threshold=thr, flags=_lib.PLAN_NO_REC)
        r2 = PlanReference.of(CsrOperator.from_scipy(m))
        r2.build_hub_plan(256, thr)
        _compare(B, r2, dev)
    C = CsrOperator.from_scipy(m, dev).build_plans(256, flags=_lib.PLAN_NO_HUB)
    assert C.hub is None and C.view().hub_n == 0


def test_handle_through_ctypes_alone_reaches_the_group_record_kernels(dev):
    """What a caller that binds only include/ndcn_hip.h gets: ndcn_csr_create on a lattice's arrays, ndcn_csr_view into
    ndcn_rhs_f32 -> the fused right-hand side runs on rhs_fused3; ndcn_csr_info reports the plan; destroy frees it."""
    from ndcn_amd import _lib, graphs
    lib = _lib.load()
    m = graphs.normalized_laplacian(graphs.grid_8_neighbor(40)).tocsr()
    m.sort_indices()
    n, H = m.shape[0], 256
    rp = torch.from_numpy(m.indptr.astype(np.int32)).to(dev)
    ci = torch.from_numpy(m.indices.astype(np.int32)).to(dev)
    va = torch.from_numpy(m.data.astype(np.float32)).to(dev)
    h = ctypes.c_void_p()
    _lib.check(lib.ndcn_csr_create(n, n, m.nnz, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), H, None, _lib.stream_ptr(), ctypes.byref(h)))
    try:
        info = (ctypes.c_int64 * 16)()
        _lib.check(lib.ndcn_csr_info(h, info))
        assert (info[0], info[1], info[2]) == (16, 40, 2) and info[6] == 40 and info[4] == m.nnz and info[14] == 1
        torch.manual_seed(0)
        W, b = torch.randn(H, H, device=dev) / 16, torch.randn(H, device=dev)
        X = torch.rand(n, H, device=dev)
        Y = torch.empty_like(X)
        work = torch.empty(int(lib.ndcn_rhs_work_bytes(n, H, _lib.F_RELU)), dtype=torch.uint8, device=dev)
        _lib.check(lib.ndcn_rhs_f32(lib.ndcn_csr_view(h), X.data_ptr(), None, n, W.data_ptr(), b.data_ptr(), Y.data_ptr(),
                                    work.data_ptr(), H, _lib.F_RELU, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert lib.ndcn_debug_last_rhs_path() == _lib.PATH_FUSED3
        S = torch.from_numpy(m.astype(np.float64) @ X.cpu().double().numpy())
        ref = torch.relu(S @ W.cpu().double().T + b.cpu().double())
        assert float((Y.cpu().double() - ref).abs().max()) < 2e-5
    finally:
        _lib.check(lib.ndcn_csr_destroy(h))
    # argument errors come back as codes
    bad = ctypes.c_void_p()
    assert lib.ndcn_csr_create(n, n, m.nnz, None, ci.data_ptr(), va.data_ptr(), H, None, None, ctypes.byref(bad)) == _lib.EINVAL
    hints = _lib.CsrHints()
    hints.rec_rows, hints.rec_cap, hints.rec_kib = 8, 33, 1
    assert lib.ndcn_csr_create(n, n, m.nnz, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), H, ctypes.byref(hints), None, ctypes.byref(bad)) == _lib.EINVAL
