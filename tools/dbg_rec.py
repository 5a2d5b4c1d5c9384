import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch, scipy.sparse as sp
from ndcn_amd import hip, CsrOperator, graphs
dev = torch.device('cuda:0')
side, H = 41, 256
n = side * side
grid = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
def rand_csr(n_rows, n_cols, avg, seed):
    rng = np.random.RandomState(seed)
    deg = rng.poisson(avg, size=n_rows)
    rows = np.repeat(np.arange(n_rows), deg)
    cols = rng.randint(0, n_cols, size=rows.size)
    m = sp.csr_matrix((rng.randn(rows.size).astype(np.float32), (rows, cols)), shape=(n_rows, n_cols))
    m.sum_duplicates(); m.sort_indices(); return m
rnd = rand_csr(n, n, 7, 9)
for name, m in (('grid', grid), ('mixed', sp.vstack([grid[:800], rnd[800:]]).tocsr())):
    m.sort_indices()
    for shape in ((8, 32, 1), (16, 40, 2)):
        A = CsrOperator.from_scipy(m, dev); A._plans_tried = True
        A.group_order = torch.as_tensor(CsrOperator.from_scipy(grid, dev).detect_stencil_order(), dtype=torch.int32).to(dev)
        print(name, shape, A.build_rec_plan(*shape))
        g = torch.Generator().manual_seed(2)
        X, y0 = torch.rand(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev)
        ks = [torch.randn(n, H, generator=g).to(dev) for _ in range(5)]
        cs = [np.float32(c) for c in (0.11, -0.07, 0.23, 0.05, -0.31, 0.19)]
        P = CsrOperator.from_scipy(m, dev); P._plans_tried = True
        K_ref = hip.rhs(P, X, None, None, no_control=True)
        s2, b2 = hip.error(y0, X, ks + [K_ref], cs, 1e-2, 1e-3)
        for rep in range(4):
            for npv in (5, 3, 0):
                K, (s1, b1) = hip.rhs_rk(A, X, None, None, 'error', y0, ks[:npv], cs[:npv] + [cs[5]], rtol=1e-2, atol=1e-3, no_control=True)
                s3, b3 = hip.error(y0, X, ks[:npv] + [K_ref], cs[:npv] + [cs[5]], 1e-2, 1e-3)
                print('  rep', rep, 'np', npv, 'fused', s1, b1, 'ref', s3, b3, 'Kequal', bool(torch.equal(K, K_ref)))
