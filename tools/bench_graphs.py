#!/usr/bin/env python3
"""The bench's dopri5 step on the other graph families of BASELINE.json's configs (parity-test cases, not bench
lines): how the fused RHS kernel copes with skewed degree distributions.  One JSON line per graph."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n', type=int, default=1000000)
    p.add_argument('--hidden', type=int, default=256)
    p.add_argument('--steps', type=int, default=12)
    p.add_argument('--networks', default='grid,power_law,small_world,random')
    a = p.parse_args()
    from ndcn_amd import graphs, _lib, CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    import bench
    dev = torch.device('cuda:0')
    lib = _lib.load()
    for net in a.networks.split(','):
        t0 = time.time()
        G = graphs.make_graph(net, a.n, seed=0)
        L = graphs.normalized_laplacian(G)
        deg = np.diff(L.indptr)
        A = CsrOperator.from_scipy(L, dev)
        torch.manual_seed(0)
        f = ODEFunc(a.hidden, A).to(dev)
        x0 = torch.rand(L.shape[0], a.hidden, generator=torch.Generator().manual_seed(0)).to(dev)
        build_s = time.time() - t0
        with torch.no_grad():
            r = bench.SingleGpuRunner(f, x0, 5.0, 0.01, 0.001)
            r.run_steps(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r.run_steps(a.steps)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t1
            nk = lib.ndcn_prof_kinds()
            buf = (_lib.ctypes.c_double * (4 * nk))()
            lib.ndcn_prof_enable(1)
            lib.ndcn_prof_read(buf, nk)
            r.run_steps(a.steps)
            torch.cuda.synchronize()
            lib.ndcn_prof_enable(0)
            lib.ndcn_prof_read(buf, nk)
        i = _lib.PROF_KINDS.index('rhs_fused')
        cnt, ms, byt, fl = buf[4 * i:4 * i + 4]
        kinds = {k: [int(buf[4 * j]), round(buf[4 * j + 1], 2)] for j, k in enumerate(_lib.PROF_KINDS) if buf[4 * j]}
        print(json.dumps({'network': net, 'n': int(L.shape[0]), 'nnz': int(L.nnz), 'max_degree': int(deg.max()),
                          'rows_over_64': int((deg > 64).sum()), 'ms_per_step': round(1e3 * wall / a.steps, 3),
                          'rhs_fused_avg_ms': round(ms / max(cnt, 1), 4), 'rhs_fused_GBps': round(byt / max(ms, 1e-9) / 1e6, 1),
                          'rhs_fused_TFLOPs': round(fl / max(ms, 1e-9) / 1e9, 1), 'graph_build_s': round(build_s, 1),
                          'kernels_launches_ms': kinds, 'hub_rows': 0 if A.hub is None else A.hub['n'],
                          'hub_nnz': 0 if A.hub is None else A.hub['nnz']}), flush=True)
        del r, f, x0, A


if __name__ == '__main__':
    main()
