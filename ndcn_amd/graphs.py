"""O(nnz) graph generators and operator builders for the hot path's inputs.

The reference builds every graph and operator as a DENSE N x N matrix first
(utils_in_learn_dynamics.py:80-157; heat_dynamics.py:80-117,150-175), which caps it at ~10^4 nodes.
These builders produce the same matrices directly in scipy CSR (host side, numpy) so the 10^5..4*10^6
node configurations of BASELINE.json exist at all; `tests/test_graphs.py` pins them to the reference's
dense results at N = 400 (bit-exact values for the operators).

Rounding notes (so the sparse values equal the dense ones bit for bit):
  normalized_laplacian / normalized_adj : degrees cast to float32, d^-1/2 in float32, the two scalings are
      float32 products fl(fl(d_i a_ij) d_j)                       (utils_in_learn_dynamics.py:114-119,128-133)
  zipf_smoothing : A + I is float64 (np.eye), so the scalings are float64 products, rounded to float32 when
      the driver wraps the result in torch.FloatTensor              (utils_in_learn_dynamics.py:85-91)
  zipf_smoothing_alpha : float64 products on scipy matrices, cast to float32 by the loader
                                                                    (propagation.py:91-103; utils.py:18)
  zero-degree nodes: the reference's np.power(..., where=deg != 0) leaves those slots uninitialised;
      here they are 0.
"""
import numpy as np
import scipy.sparse as sp

from .csr import CsrOperator


# ---------------------------------------------------------------------------------------------------
# graphs (symmetric 0/1 adjacency, no self loops, scipy CSR float32)
# ---------------------------------------------------------------------------------------------------

def _sym_csr(u, v, n):
    """Undirected simple graph from an edge list: drop self loops and duplicates, symmetrise."""
    u = np.asarray(u, dtype=np.int64)
    v = np.asarray(v, dtype=np.int64)
    keep = u != v
    u, v = u[keep], v[keep]
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    rows = np.concatenate([lo, hi])
    cols = np.concatenate([hi, lo])
    m = sp.csr_matrix((np.ones(rows.size, dtype=np.float32), (rows, cols)), shape=(n, n))
    m.sort_indices()
    return m


def grid_8_neighbor(S):
    """S x S lattice, 8 neighbours, no wrap-around; node (x, y) -> x*S + y
    (utils_in_learn_dynamics.py:137-157)."""
    S = int(S)
    idx = np.arange(S * S, dtype=np.int64).reshape(S, S)
    us, vs = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            xs = slice(max(0, -dx), S - max(0, dx))
            ys = slice(max(0, -dy), S - max(0, dy))
            xd = slice(max(0, dx), S - max(0, -dx))
            yd = slice(max(0, dy), S - max(0, -dy))
            us.append(idx[xs, ys].ravel())
            vs.append(idx[xd, yd].ravel())
    u, v = np.concatenate(us), np.concatenate(vs)
    m = sp.csr_matrix((np.ones(u.size, dtype=np.float32), (u, v)), shape=(S * S, S * S))
    m.sort_indices()
    return m


def grid_8_neighbor_rect(R, C):
    """R x C lattice, 8 neighbours, node (x, y) -> x*C + y (the square case is grid_8_neighbor)."""
    R, C = int(R), int(C)
    idx = np.arange(R * C, dtype=np.int64).reshape(R, C)
    us, vs = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            us.append(idx[max(0, -dx):R - max(0, dx), max(0, -dy):C - max(0, dy)].ravel())
            vs.append(idx[max(0, dx):R - max(0, -dx), max(0, dy):C - max(0, -dy)].ravel())
    u, v = np.concatenate(us), np.concatenate(vs)
    m = sp.csr_matrix((np.ones(u.size, dtype=np.float32), (u, v)), shape=(R * C, R * C))
    m.sort_indices()
    return m


def grid_operator_row_block(R, C, x0, x1, kind='norm_lap'):
    """Rows [x0*C, x1*C) of the operator of the R x C grid, with GLOBAL column ids, built from a window of
    lattice rows [x0-2, x1+2) only (degrees of every node that appears in those rows are exact), so a
    rank can build its shard of a grid that is too large to build whole."""
    lo, hi = max(0, x0 - 2), min(R, x1 + 2)
    sub = make_operator(grid_8_neighbor_rect(hi - lo, C), kind).tocsr()
    block = sub[(x0 - lo) * C:(x1 - lo) * C]
    block = sp.csr_matrix((block.data, block.indices.astype(np.int64) + lo * C, block.indptr),
                          shape=(block.shape[0], R * C))
    return block


def erdos_renyi(n, p, seed=0):
    """G(n, p) (heat_dynamics.py:89) sampled in O(expected edges): draw the edge count, then distinct pairs."""
    rng = np.random.RandomState(seed)
    total = n * (n - 1) // 2
    m = rng.binomial(total, p) if total < 2 ** 62 else int(total * p)
    got = np.empty(0, dtype=np.int64)
    while got.size < m:
        need = int((m - got.size) * 1.1) + 16
        u = rng.randint(0, n, size=need, dtype=np.int64)
        v = rng.randint(0, n, size=need, dtype=np.int64)
        ok = u != v
        lo, hi = np.minimum(u[ok], v[ok]), np.maximum(u[ok], v[ok])
        got = np.unique(np.concatenate([got, lo * n + hi]))
    got = rng.permutation(got)[:m]
    return _sym_csr(got // n, got % n, n)


def barabasi_albert(n, m, seed=0):
    """Preferential attachment, m edges per new node (heat_dynamics.py:94: m = 5)."""
    rng = np.random.RandomState(seed)
    # repeated-endpoints list: a node appears once per incident edge, so uniform draws are degree-proportional
    rep = np.empty(2 * m * n, dtype=np.int64)
    fill = 0
    src = np.empty(m * (n - m), dtype=np.int64)
    dst = np.empty(m * (n - m), dtype=np.int64)
    targets = np.arange(m, dtype=np.int64)
    e = 0
    for new in range(m, n):
        src[e:e + m] = new
        dst[e:e + m] = targets
        e += m
        rep[fill:fill + m] = targets
        rep[fill + m:fill + 2 * m] = new
        fill += 2 * m
        # m distinct degree-proportional targets for the next node
        chosen = set()
        while len(chosen) < m:
            for c in rep[rng.randint(0, fill, size=m - len(chosen))]:
                chosen.add(int(c))
        targets = np.fromiter(chosen, dtype=np.int64, count=m)
    return _sym_csr(src, dst, n)


def newman_watts_strogatz(n, k, p, seed=0):
    """Ring of k//2 neighbours per side plus, per ring edge, a shortcut with probability p
    (heat_dynamics.py:99: k = 5, p = 0.5)."""
    rng = np.random.RandomState(seed)
    half = k // 2
    base = np.arange(n, dtype=np.int64)
    us, vs = [], []
    for j in range(1, half + 1):
        us.append(base)
        vs.append((base + j) % n)
    u, v = np.concatenate(us), np.concatenate(vs)
    add = rng.random_sample(u.size) < p
    w = rng.randint(0, n, size=int(add.sum()), dtype=np.int64)
    return _sym_csr(np.concatenate([u, u[add]]), np.concatenate([v, w]), n)


def random_partition(sizes, p_in, p_out, seed=0):
    """Planted-partition graph (heat_dynamics.py:104-108), sampled block pair by block pair."""
    rng = np.random.RandomState(seed)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(starts[-1])
    us, vs = [], []
    for a in range(len(sizes)):
        for b in range(a, len(sizes)):
            na, nb = int(sizes[a]), int(sizes[b])
            pairs = na * (na - 1) // 2 if a == b else na * nb
            cnt = rng.binomial(pairs, p_in if a == b else p_out)
            # with-replacement draws; duplicates are collapsed by _sym_csr (negligible at these densities)
            u = rng.randint(0, na, size=cnt, dtype=np.int64) + starts[a]
            v = rng.randint(0, nb, size=cnt, dtype=np.int64) + starts[b]
            us.append(u)
            vs.append(v)
    return _sym_csr(np.concatenate(us), np.concatenate(vs), n)


def make_graph(network, n, seed=0, mean_degree=None, layout=None):
    """The five `--network` choices of the drivers (heat_dynamics.py:83-110) at arbitrary n; every network but the
    grid is re-labelled by `layout` ('degree' / 'community' / None) as the reference does (:90,95,100,109).

    The reference's densities are quoted for n = 400 (ER p = 0.1, partition p = .25 / .01); at scale the
    same MEAN DEGREE is kept instead unless `mean_degree` is given."""
    if network == 'grid':
        S = int(np.ceil(np.sqrt(n)))
        return grid_8_neighbor(S)
    if network == 'random':
        deg = 39.9 if mean_degree is None else mean_degree
        G = erdos_renyi(n, min(1.0, deg / max(n - 1, 1)), seed)
    elif network == 'power_law':
        G = barabasi_albert(n, 5, seed)
    elif network == 'small_world':
        G = newman_watts_strogatz(n, 5, 0.5, seed)
    elif network == 'community':
        n1, n2, n3 = int(n / 3), int(n / 3), int(n / 4)
        sizes = [n1, n2, n3, n - n1 - n2 - n3]
        scale = 400.0 / n
        G = random_partition(sizes, min(1.0, .25 * scale), min(1.0, .01 * scale), seed)
    else:
        raise ValueError('unknown network %r' % network)
    return reorder_nodes(G, layout)


# ---------------------------------------------------------------------------------------------------
# node layouts (the drivers' --layout)
# ---------------------------------------------------------------------------------------------------

def node_mapping(A, layout):
    """new_index[old node] for the drivers' `--layout` choices (utils_in_learn_dynamics.py:212-230).

    degree    : nodes ranked by degree, largest first; ties keep their original order (the reference's
                `sorted(G.degree, key=deg, reverse=True)` is stable).  O(n log n) on the CSR row lengths.
    community : the communities of networkx's greedy_modularity_communities (the reference's own dependency),
                largest first, members in the order `list(frozenset)` yields - what the reference executes.  The
                algorithm is networkx's pure-Python one: usable up to ~10^4..10^5 nodes, like the reference.
    None      : identity (returns None)."""
    if layout is None:
        return None
    A = A.tocsr()
    n = A.shape[0]
    if layout == 'degree':
        deg = np.diff(A.indptr) + (A.diagonal() != 0)            # networkx counts a self loop twice
        order = np.argsort(-deg.astype(np.int64), kind='stable')
    elif layout == 'community':
        import networkx as nx
        from networkx.algorithms import community
        G = nx.Graph()
        G.add_nodes_from(range(n))
        coo = A.tocoo()
        keep = coo.row <= coo.col
        G.add_edges_from(zip(coo.row[keep].tolist(), coo.col[keep].tolist()))
        order = np.fromiter((v for c in community.greedy_modularity_communities(G) for v in list(c)), dtype=np.int64, count=n)
    else:
        raise ValueError('unknown layout %r' % (layout,))
    new = np.empty(n, dtype=np.int64)
    new[order] = np.arange(n, dtype=np.int64)
    return new


def reorder_nodes(A, layout):
    """P A P^T for the node mapping of `layout`: new_A[map[i], map[j]] = A[i, j]
    (networkx_reorder_nodes, utils_in_learn_dynamics.py:233-247).  O(nnz)."""
    new = node_mapping(A, layout)
    return A if new is None else permute_nodes(A, new)


def permute_nodes(A, new_index):
    """The matrix re-labelled by new_index[old] (rows and columns), as sorted CSR."""
    new_index = np.asarray(new_index, dtype=np.int64)
    coo = A.tocoo()
    m = sp.csr_matrix((coo.data, (new_index[coo.row], new_index[coo.col])), shape=A.shape)
    m.sort_indices()
    return m


# ---------------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------------

def _inv_sqrt(deg):
    deg = np.asarray(deg, dtype=np.float32).reshape(-1)
    out = np.zeros_like(deg)
    nz = deg != 0
    out[nz] = np.power(deg[nz], np.float32(-0.5))
    return out


def _coo_parts(A):
    A = A.tocsr()
    A.sort_indices()
    rows = np.repeat(np.arange(A.shape[0], dtype=np.int64), np.diff(A.indptr))
    return A, rows, A.indices.astype(np.int64)


def laplacian(A):
    """L = D - A (heat_dynamics.py:116-117)."""
    A = A.tocsr().astype(np.float32)
    deg = np.asarray(A.sum(1)).reshape(-1).astype(np.float32)
    L = (sp.diags(deg) - A).tocsr().astype(np.float32)
    L.sort_indices()
    return L


def normalized_adj(A):
    """D^-1/2 A D^-1/2 (utils_in_learn_dynamics.py:123-134), float32 products."""
    A, rows, cols = _coo_parts(A)
    do = _inv_sqrt(A.sum(1))
    di = _inv_sqrt(A.sum(0))
    vals = (do[rows] * A.data.astype(np.float32)).astype(np.float32) * di[cols]
    m = sp.csr_matrix((vals.astype(np.float32), A.indices, A.indptr), shape=A.shape)
    return m


def normalized_laplacian(A):
    """I - D^-1/2 A D^-1/2 (utils_in_learn_dynamics.py:109-120); the drivers' default operator."""
    na = normalized_adj(A)
    n = A.shape[0]
    m = (sp.identity(n, dtype=np.float64, format='csr') - na.astype(np.float64)).tocsr().astype(np.float32)
    m.sort_indices()
    return m


def zipf_smoothing(A):
    """(D+I)^-1/2 (A+I) (D+I)^-1/2 (utils_in_learn_dynamics.py:80-92), float64 products."""
    n = A.shape[0]
    Ap = (A.astype(np.float64) + sp.identity(n, dtype=np.float64, format='csr')).tocsr()
    Ap, rows, cols = _coo_parts(Ap)
    do = _inv_sqrt(Ap.sum(1)).astype(np.float64)
    di = _inv_sqrt(Ap.sum(0)).astype(np.float64)
    vals = (do[rows] * Ap.data) * di[cols]
    return sp.csr_matrix((vals.astype(np.float32), Ap.indices, Ap.indptr), shape=Ap.shape)


def zipf_smoothing_alpha(A, alpha=0.5):
    """(aI + (1-a)D)^-1/2 (aI + (1-a)A) (aI + (1-a)D)^-1/2 (propagation.py:91-103)."""
    n = A.shape[0]
    Ap = (alpha * sp.identity(n, dtype=np.float64, format='csr') + (1 - alpha) * A.astype(np.float64)).tocsr()
    Ap, rows, cols = _coo_parts(Ap)
    do = _inv_sqrt(Ap.sum(1)).astype(np.float64)
    di = _inv_sqrt(Ap.sum(0)).astype(np.float64)
    vals = (do[rows] * Ap.data) * di[cols]
    return sp.csr_matrix((vals.astype(np.float32), Ap.indices, Ap.indptr), shape=Ap.shape)


def make_operator(A, kind='norm_lap'):
    """The drivers' `--operator` choices (heat_dynamics.py:150-161)."""
    if kind == 'lap':
        return laplacian(A)
    if kind == 'kipf':
        return zipf_smoothing(A)
    if kind == 'norm_adj':
        return normalized_adj(A)
    if kind == 'norm_lap':
        return normalized_laplacian(A)
    raise ValueError('unknown operator %r' % kind)


def x0_blocks(S):
    """The drivers' initial image: three constant blocks 25 / 20 / 17 on the S x S canvas, flattened to
    N x 1 (heat_dynamics.py:178-182)."""
    S = int(S)
    x0 = np.zeros((S, S), dtype=np.float32)
    x0[int(0.05 * S):int(0.25 * S), int(0.05 * S):int(0.25 * S)] = 25
    x0[int(0.45 * S):int(0.75 * S), int(0.45 * S):int(0.75 * S)] = 20
    x0[int(0.05 * S):int(0.25 * S), int(0.35 * S):int(0.65 * S)] = 17
    return x0.reshape(-1, 1)


def grid_tile_order(R, C, tile_cols=256, n_chunks=8):
    """Walk order for an R x C lattice operator: the rows are cut into `n_chunks` contiguous lattice-row bands
    (one per XCD, matching the kernels' contiguous chunking of the order array) and each band is walked in
    column strips of `tile_cols`, lattice row by lattice row inside a strip.  A node's 8 neighbours then recur
    within ~2 (tile_cols + 2) walked rows instead of ~2 C, which keeps the X window inside one XCD's L2."""
    R, C = int(R), int(C)
    n = R * C
    per = (n + n_chunks - 1) // n_chunks          # the kernels chunk POSITIONS evenly; keep bands aligned to that
    order = np.empty(n, dtype=np.int32)
    idx = np.arange(n, dtype=np.int64)
    x, y = idx // C, idx % C
    band = idx // per
    key = (band * ((C + tile_cols - 1) // tile_cols + 1) + y // tile_cols) * (R + 1) + x
    order[:] = np.lexsort((y, key)).astype(np.int32)
    return order


def to_device(m, device):
    return CsrOperator.from_scipy(m, device)


def spmm_bytes(n_rows, nnz, H):
    """Algorithmic HBM bytes of one CSR SpMM: fp32 values + int32 indices, X read once,
    Y written once."""
    return 8 * nnz + 4 * (n_rows + 1) + 8 * n_rows * H
