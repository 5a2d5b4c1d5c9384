"""Column sweep against the row gather on G(n,p) graphs of several sizes (mean degree 40, H = 256): where the plan is taken by
itself, how multi-pass operators (n > 100 352) behave.  HIP-event timed, bit-compared."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndcn_amd import graphs, hip, CsrOperator

dev = torch.device('cuda:0')
for n in [int(a) for a in sys.argv[1:]] or [20000, 50000, 100000, 150000, 200000, 300000]:
    m = graphs.normalized_laplacian(graphs.make_graph('random', n, seed=0)).tocsr()
    m.sort_indices()
    A = CsrOperator.from_scipy(m, dev).ensure_plans(256)
    auto = A.sweep is not None
    if not auto:                                            # not taken by itself: force it, to see what the decision leaves on the table
        from ndcn_amd import _lib
        A = CsrOperator.from_scipy(m, dev).build_plans(256, flags=_lib.PLAN_FORCE_SWEEP)
        A._plans_tried = True
    R = CsrOperator.from_scipy(m, dev)
    R._plans_tried = True
    X = torch.rand(n, 256, device=dev)
    out = {}
    for name, op in (('sweep' if auto else 'sweep (FORCED)', A), ('row gather', R)):
        Y = torch.empty_like(X)
        for _ in range(3):
            hip.spmm(op, X, out=Y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.spmm(op, X, out=Y)
        e1.record()
        torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) / 20, Y)
    (na, (ta, Ya)), (nr, (tr, Yr)) = out.items()
    same = torch.equal(Ya.view(torch.int32), Yr.view(torch.int32))
    print('n = %7d  nnz = %9d  passes %s  %-16s %.3f ms   row gather %.3f ms   x%.2f   bits equal: %s'
          % (n, m.nnz, A.sweep['passes'] if A.sweep else '-', na, ta, tr, tr / ta, same), flush=True)
