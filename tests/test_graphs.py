"""O(nnz) graph / operator builders (ndcn_amd/graphs.py) against the reference's dense builders
(fixtures operators_*.npz) and structural properties at scale.  CPU-only."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from ndcn_amd import graphs


def csr_from(d, prefix, n):
    return sp.csr_matrix((d[prefix + '_data'] if prefix + '_data' in d else np.ones(len(d[prefix + '_indices']), np.float32),
                          d[prefix + '_indices'], d[prefix + '_indptr']), shape=(n, n))


def test_grid_and_operators_equal_reference_bitwise():
    d = load_golden('operators_grid400')
    A = graphs.grid_8_neighbor(20)
    ref_A = csr_from(d, 'A', 400)
    assert (A != ref_A).nnz == 0
    for kind in ('norm_lap', 'norm_adj', 'kipf', 'lap'):
        op = graphs.make_operator(A, kind)
        ref = csr_from(d, kind, 400)
        op.eliminate_zeros()
        assert np.array_equal(op.indptr, ref.indptr) and np.array_equal(op.indices, ref.indices), kind
        assert np.array_equal(op.data.astype(np.float32), ref.data), kind     # bit-exact values
    assert np.array_equal(graphs.x0_blocks(20), d['x0'])
    d5 = load_golden('operators_grid25')
    assert (graphs.grid_8_neighbor(5) != csr_from(d5, 'A', 25)).nnz == 0


@pytest.mark.parametrize('name', ['cora', 'pubmed'])
def test_zipf_alpha_equals_reference(name):
    d = load_golden('operators_' + name)
    n = int(d['n'])
    adj = csr_from(d, 'adj', n)
    for alpha, tag in ((0.0, 'alpha00'), (0.5, 'alpha05')):
        op = graphs.zipf_smoothing_alpha(adj, alpha)
        ref = csr_from(d, tag, n)
        assert abs(op - ref).max() <= 1e-7


def test_grid_nnz_formula_and_scale():
    # nnz(norm_lap) = 4 (S-1)(2S-1) + S^2 ; 1000 x 1000 -> 8 988 004
    A = graphs.grid_8_neighbor(300)
    assert A.nnz == 4 * 299 * 599
    L = graphs.normalized_laplacian(A)
    assert L.nnz == 4 * 299 * 599 + 300 * 300
    assert abs(L - L.T).max() < 1e-7
    assert np.allclose(L.diagonal(), 1.0)


@pytest.mark.parametrize('network,lo,hi', [('random', 38, 42), ('power_law', 9.5, 10.1), ('small_world', 5.5, 6.1),
                                           ('community', 20, 45)])
def test_generators_structure(network, lo, hi):
    n = 20000
    A = graphs.make_graph(network, n, seed=1)
    assert A.shape == (n, n)
    assert (A != A.T).nnz == 0                     # undirected
    assert A.diagonal().sum() == 0                 # no self loops
    assert set(np.unique(A.data)) == {1.0}
    assert lo <= A.nnz / n <= hi, A.nnz / n
    B = graphs.make_graph(network, n, seed=1)
    assert (A != B).nnz == 0                       # seeded


def test_zero_degree_nodes_are_defined():
    A = sp.csr_matrix(np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]], dtype=np.float32))
    for kind in ('norm_lap', 'norm_adj', 'kipf', 'lap'):
        op = graphs.make_operator(A, kind).toarray()
        assert np.isfinite(op).all()
    assert graphs.normalized_adj(A).toarray()[2].sum() == 0


def test_algorithmic_bytes_of_the_metric_case():
    # 1M-node grid, H=256 -> 2.124 GB per SpMM
    assert abs(graphs.spmm_bytes(10 ** 6, 8988004, 256) / 1e9 - 2.124) < 1e-3


def _decode_rec(A):
    """Rebuild (row -> columns, values) from a group-record plan; returns (rows seen, staged entries)."""
    rec = A.rec['rec'].numpy()
    R, CAP = A.rec['rows'], A.rec['cap']
    E0 = CAP + 2 * R
    rp, ci, va = A.rowptr.numpy(), A.colidx.numpy(), A.val.numpy()
    seen, staged = [], 0
    for g in range(rec.shape[0]):
        r = rec[g]
        cols = r[:CAP]
        for i in range(R):
            row, meta = int(r[CAP + 2 * i]), int(r[CAP + 2 * i + 1])
            if row < 0:
                continue
            seen.append(row)
            cnt, ofs = meta & 0xffff, meta >> 16
            if cnt == 0xffff:                                   # group the record cannot hold: gathered from the CSR
                continue
            assert cnt == rp[row + 1] - rp[row] <= 64
            e = r[E0 + 2 * ofs:E0 + 2 * (ofs + cnt)].reshape(-1, 2)
            assert np.array_equal(cols[e[:, 0]], ci[rp[row]:rp[row + 1]])
            assert np.array_equal(e[:, 1].view(np.float32), va[rp[row]:rp[row + 1]])
            staged += cnt
        live = np.unique(cols)
        assert np.all(np.diff(cols[:live.size]) > 0) or live.size <= 1     # ascending, duplicate-free, then padding
    return seen, staged


def test_group_record_plan_reconstructs_the_operator():
    """The plan restatement the GPU plan tests compare the library with (tests/_plan_reference.py; the layout of
    include/ndcn_hip.h, ndcn_csr::rec): every row appears once, staged rows decode to their CSR entries, groups that do
    not fit are flagged; lattice detection yields whole patches."""
    import torch
    from ndcn_amd import CsrOperator as _Csr
    from _plan_reference import PlanReference
    import scipy.sparse as sp

    class CsrOperator:
        REC_SHAPES = _Csr.REC_SHAPES

        @staticmethod
        def from_scipy(m):
            return PlanReference.of(_Csr.from_scipy(m))
    rng = np.random.RandomState(0)
    grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(37))          # side not a multiple of the patch
    rnd = sp.random(1369, 1369, density=0.01, random_state=rng, format='csr', dtype=np.float32)
    longrow = sp.vstack([grid[:500], sp.csr_matrix(np.ones((1, 1369), np.float32)), grid[501:]]).tocsr()
    mixed = sp.vstack([grid[:800], rnd[800:]]).tocsr()
    for m, hinted in ((grid, True), (grid, False), (mixed, False), (rnd, False), (longrow, True)):
        m.sort_indices()
        for shape in CsrOperator.REC_SHAPES:
            A = CsrOperator.from_scipy(m)
            if hinted:
                order = CsrOperator.from_scipy(grid).detect_stencil_order()
                assert order is not None and order.size % 16 == 0 and sorted(order[order >= 0]) == list(range(1369))
                A.group_order = torch.as_tensor(order)
            frac, loads = A.build_rec_plan(*shape)
            seen, staged = _decode_rec(A)
            assert sorted(seen) == list(range(m.shape[0]))
            assert abs(staged / m.nnz - frac) < 1e-9
            if m is grid and (hinted or shape[0] == 8):
                assert frac == 1.0 and loads < (2.3 if hinted and shape[0] == 16 else 4.1)
            if m is rnd:
                assert frac < 0.2
            if m is longrow:
                assert frac < 1.0                                   # the 1369-entry row's group is gathered directly
    assert CsrOperator.from_scipy(rnd).detect_stencil_order() is None
    five = sp.diags([1., 1., 1., 1., 1.], [-50, -1, 0, 1, 50], shape=(2500, 2500), format='csr', dtype=np.float32)
    assert CsrOperator.from_scipy(five).detect_stencil_order() is not None    # 5-point stencil, stride 50


@pytest.mark.parametrize('shape,avg', [((3000, 3000), 12), ((700, 5000), 9), ((100352 + 5, 600), 2)])
def test_column_sweep_plan_restatement_reconstructs_the_operator(shape, avg):
    """tests/_plan_reference.py: sweep_plan_reference - what tests/test_gpu_sweep.py compares the library's column-sweep plan with -
    decodes back to the operator: every entry once, in its slab, the slab's stream sorted by column (ties in row order), rows in
    ascending column order, padding to groups of 8 with the dummy row, the table pointing at group-aligned starts."""
    from _plan_reference import sweep_plan_reference
    rng = np.random.RandomState(shape[0])
    deg = rng.poisson(avg, size=shape[0])
    deg[:3] = 0
    rows = np.repeat(np.arange(shape[0]), deg)
    m = sp.csr_matrix((rng.randn(rows.size).astype(np.float32), (rows, rng.randint(0, shape[1], size=rows.size))), shape=shape)
    m.sum_duplicates()
    m.sort_indices()
    ref = sweep_plan_reference(m.indptr, m.indices, m.data, m.shape)
    n = shape[0]
    assert ref['passes'] == (n + 100351) // 100352 and ref['slab'].shape == (ref['passes'] * 2048, 2)
    assert ref['rpw'] <= 49 and ref['ent'].shape == (ref['entries'] + 32, 2)
    assert (ref['slab'][:, 0] % 8 == 0).all() and int(ref['slab'][:, 1].sum()) == m.nnz
    rr, cc, vv = [], [], []
    used = np.zeros(ref['entries'] + 32, bool)
    for s_, (start, cnt) in enumerate(ref['slab']):
        e = ref['ent'][start:start + cnt]
        used[start:start + cnt] = True
        if not cnt:
            continue
        col = (e[:, 0] & 0xffffff).astype(np.int64)
        loc = (e[:, 0] >> 24).astype(np.int64)
        assert (np.diff(col) >= 0).all() and loc.max() < ref['rpw']
        tie = np.diff(col) == 0
        assert (np.diff(loc)[tie] > 0).all()                              # equal columns: in row order (a stable sort)
        p_, x_, sl_ = s_ // 2048, (s_ % 2048) // 256, s_ % 256
        base = p_ * ref['rows_per_pass']
        end = min(n, base + ref['rows_per_pass'])
        per_xcd = (end - base + 7) // 8
        rpw = (per_xcd + 255) // 256
        rr.append(base + x_ * per_xcd + sl_ * rpw + loc)
        cc.append(col)
        vv.append(e[:, 1].copy().view(np.float32))
    assert (ref['ent'][~used] == np.array([49 << 24, 0], dtype=np.uint32)).all()      # padding: 0 * X[0] into the dummy row
    back = sp.csr_matrix((np.concatenate(vv), (np.concatenate(rr), np.concatenate(cc))), shape=shape)
    back.sort_indices()
    assert np.array_equal(back.indptr, m.indptr) and np.array_equal(back.indices, m.indices) and np.array_equal(back.data, m.data)


def test_planetoid_loader_against_reference_steps():
    """ndcn_amd/planetoid.py vs the fixture assembled from the reference's data files (tools/gen_golden.py G8).
    Needs the reference's data directory; skipped where it is absent (GPU box)."""
    import os
    import pytest
    import torch
    ref_data = '/root/reference/data'
    if not os.path.isdir(ref_data):
        pytest.skip('reference data directory not present')
    from ndcn_amd import planetoid
    d = load_golden('dataset_cora')
    adj, feats, labels, itr, iva, ite = planetoid.load_data('cora', 0.0, ref_data)
    ref_feat = sp.csr_matrix((d['feat_data'], d['feat_indices'].astype(np.int64), d['feat_indptr']), shape=tuple(d['feat_shape']))
    assert np.abs(feats.numpy() - ref_feat.toarray()).max() < 1e-7
    assert np.array_equal(labels.numpy(), d['labels'].astype(np.int64))
    assert np.array_equal(itr.numpy(), d['idx_train']) and np.array_equal(iva.numpy(), d['idx_val'])
    assert np.array_equal(ite.numpy(), d['idx_test'])
    g = load_golden('operators_cora')
    ref_op = sp.csr_matrix((g['alpha00_data'], g['alpha00_indices'], g['alpha00_indptr']), shape=(2708, 2708))
    assert abs(adj.to_scipy() - ref_op).max() < 1e-7


@pytest.mark.parametrize('net', ['random', 'power_law', 'small_world', 'community'])
@pytest.mark.parametrize('layout', ['degree', 'community'])
def test_node_layout_equals_reference_mapping(golden, net, layout):
    """--layout (utils_in_learn_dynamics.py:212-247): fixture = the reference's generate_node_mapping on its own
    networkx graphs at n = 400 and the re-labelled adjacency (tools/gen_golden.py gen_layout)."""
    g = golden('layout_' + net)
    n = g['indptr'].size - 1
    A = sp.csr_matrix((np.ones(g['indices'].size, np.float32), g['indices'], g['indptr']), shape=(n, n))
    new = graphs.node_mapping(A, layout)
    assert np.array_equal(new, g['map_' + layout])
    B = graphs.reorder_nodes(A, layout)
    assert np.array_equal(B.indptr, g['new_indptr_' + layout]) and np.array_equal(B.indices, g['new_indices_' + layout])
    # a relabelling: same degree multiset, symmetric, and undone by the inverse permutation
    inv = np.empty(n, dtype=np.int64)
    inv[new] = np.arange(n)
    back = graphs.permute_nodes(B, inv)
    assert (back != A).nnz == 0


def test_make_graph_applies_layout():
    A0 = graphs.make_graph('power_law', 300, seed=1)
    A1 = graphs.make_graph('power_law', 300, seed=1, layout='degree')
    deg = np.diff(A1.indptr)
    assert np.all(deg[:-1] >= deg[1:])                       # degree layout: non-increasing degrees
    assert sorted(np.diff(A0.indptr)) == sorted(deg)
    assert graphs.make_graph('grid', 100, layout='degree').shape == (100, 100)   # the grid is never re-labelled
