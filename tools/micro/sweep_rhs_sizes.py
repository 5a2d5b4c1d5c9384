"""The right-hand side relu(W (A X) + b) on G(n,p) graphs: column sweep + rhs_fused3 on the identity operator against the one-launch
fused kernel with its row gather (plan-free operator).  HIP-event timed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndcn_amd import graphs, hip, CsrOperator

dev = torch.device('cuda:0')
torch.manual_seed(0)
W, b = ((torch.rand(256, 256) - 0.5) / 8).to(dev), ((torch.rand(256) - 0.5) / 8).to(dev)
for n in [int(a) for a in sys.argv[1:]] or [100000, 200000, 300000]:
    m = graphs.normalized_laplacian(graphs.make_graph('random', n, seed=0)).tocsr()
    m.sort_indices()
    A = CsrOperator.from_scipy(m, dev).ensure_plans(256)
    R = CsrOperator.from_scipy(m, dev)
    R._plans_tried = True
    X = torch.rand(n, 256, device=dev)
    res = []
    with torch.no_grad():
        for op in (A, R):
            K = torch.empty_like(X)
            for _ in range(3):
                hip.rhs(op, X, W, b, out=K)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.rhs(op, X, W, b, out=K)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 20, K))
    print('n = %7d  passes %s  sweep + dense stage %.3f ms   one-launch gather %.3f ms   x%.2f   equal: %s'
          % (n, A.sweep['passes'] if A.sweep else '-', res[0][0], res[1][0], res[1][0] / res[0][0], torch.equal(res[0][1], res[1][1])), flush=True)
