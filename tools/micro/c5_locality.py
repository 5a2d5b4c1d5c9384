"""Would a locality-improving node order speed up the C5 launches (Pubmed topology, no_control, H = 256: the row SpMM with the stage
algebra in its epilogue)?  Natural (dataset) order vs reverse Cuthill-McKee vs a random order, same kernel, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scipy.sparse as sp, torch
from scipy.sparse.csgraph import reverse_cuthill_mckee
from conftest import load_golden
from ndcn_amd import CsrOperator, graphs
from ndcn_amd.ops import hip
dev = torch.device('cuda:0')
g = load_golden('operators_pubmed')
n = int(g['n']); key = 'alpha00' if 'alpha00_indptr' in g else 'op'
A = sp.csr_matrix((g[key + '_data'], g[key + '_indices'], g[key + '_indptr']), shape=(n, n))
H = 256
def timed(M, label):
    op = CsrOperator.from_arrays(M.indptr, M.indices, M.data, M.shape, dev)
    x = torch.rand(n, H, device=dev); y0 = torch.rand(n, H, device=dev)
    ks = [torch.rand(n, H, device=dev) for _ in range(4)]
    cs = [np.float32(0.1)] * 5
    for reps in (5, 50):
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(reps):
            hip.rhs_rk(op, x, None, None, 'combine', y0, ks, cs, no_control=True)
        e1.record(); torch.cuda.synchronize()
    print('%-10s nnz %d  bandwidth(mean |i-j|) %.0f   %.1f us per COMBINE launch (4 earlier stages)' % (label, M.nnz, np.abs(M.tocoo().row - M.tocoo().col).mean(), 1e3 * e0.elapsed_time(e1) / reps))
timed(A, 'natural')
p = reverse_cuthill_mckee(A, symmetric_mode=True)
new = np.empty(n, dtype=np.int64); new[p] = np.arange(n)
timed(graphs.permute_nodes(A, new), 'rcm')
rng = np.random.default_rng(0); new = rng.permutation(n)
timed(graphs.permute_nodes(A, new), 'random')
