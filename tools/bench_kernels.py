#!/usr/bin/env python3
"""Kernel microbenchmarks on the metric's case (1M-node grid, H=256) - tuning aid, not the judged bench.
Each variant runs in its own process because the library reads its tuning knobs (env) once.
    python tools/bench_kernels.py            # sweep
    python tools/bench_kernels.py --one NAME # one variant in this process
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    'spmm_rec': {},                                                # group-record plan (default on lattices)
    'spmm_rec_rows8': {'NDCN_REC_STENCIL': '0'},                   # ... without the lattice patch order
    'spmm_wide': {'NDCN_REC_PLAN': '0'},                           # direct gather, one row per wave (v_readlane broadcast)
    'spmm_blocked': {'NDCN_REC_PLAN': '0', 'NDCN_SPMM_WIDE': '0'},
    'rhs_fused': {},
    'rhs_unfused': {'NDCN_RHS_FUSED': '0'},
}


def one(name, side=1000, H=256, reps=20):
    import torch
    from ndcn_amd import graphs, hip
    dev = torch.device('cuda:0')
    n = side * side
    synth = os.environ.get('SYNTH')
    if synth:
        import numpy as np
        import scipy.sparse as sp
        i = np.arange(n, dtype=np.int64)
        if synth == 'band9':
            cols = np.clip(i[:, None] + np.arange(-4, 5)[None, :], 0, n - 1)
        elif synth == 'self9':
            cols = np.repeat(i[:, None], 9, 1)
        else:
            cols = i[:, None]
        k = cols.shape[1]
        L = sp.csr_matrix((np.full(n * k, 0.1, np.float32), cols.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    else:
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = graphs.to_device(L, dev)
    if os.environ.get('TILE'):
        A.set_row_order(graphs.grid_tile_order(side, side, int(os.environ['TILE'])))
    torch.manual_seed(0)
    X = torch.rand(n, H, device=dev)
    lin = torch.nn.Linear(H, H).to(dev)
    W, b = lin.weight.detach(), lin.bias.detach()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    res = {'variant': name}
    Y = torch.empty_like(X)
    if name.startswith('spmm'):
        ms = timeit(lambda: hip.spmm(A, X, out=Y))
        res.update(ms=ms, GBps=graphs.spmm_bytes(n, L.nnz, H) / ms / 1e6)
        if os.environ.get('NDCN_UNION_DMA'):
            plain = graphs.to_device(L, dev)
            plain._plans_tried = True
            res['equal_to_plain'] = bool(torch.equal(hip.spmm(plain, X), Y))
        if os.environ.get('TILE'):       # order hint must not change the result
            Y0 = hip.spmm(graphs.to_device(L, dev), X)
            res['equal_to_unordered'] = bool(torch.equal(Y0, Y))
    else:
        ms = timeit(lambda: hip.rhs(A, X, W, b, out=Y))
        fl = 2.0 * L.nnz * H + 2.0 * n * H * H
        res.update(ms=ms, TFLOPs=fl / ms / 1e9, GBps=graphs.spmm_bytes(n, L.nnz, H) / ms / 1e6)
        # cross-check fused vs unfused result on a row sample
        S = hip.spmm(A, X)
        ref = torch.relu(torch.nn.functional.linear(S[:4096].double(), W.double(), b.double()))
        res['max_err_vs_fp64'] = float((Y[:4096].double() - ref).abs().max())
    if name == 'rhs_unfused':
        S = torch.empty_like(X)
        res['linear_ms'] = timeit(lambda: hip.linear(X, W, b, relu=True))
        ks = [torch.rand_like(X) for _ in range(6)]
        cs = [0.1 * (i + 1) for i in range(6)]
        for nk in (1, 3, 6):
            ms = timeit(lambda: hip.combine(X, ks[:nk], cs[:nk]))
            res['combine%d_ms' % nk] = ms
            res['combine%d_GBps' % nk] = 4.0 * n * H * (nk + 2) / ms / 1e6
        ms = timeit(lambda: Y.copy_(X))
        res['copy_GBps'] = 8.0 * n * H / ms / 1e6
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if '--one' in sys.argv:
        one(sys.argv[sys.argv.index('--one') + 1])
    else:
        for name, env in VARIANTS.items():
            e = dict(os.environ)
            e.update(env)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--one', name], env=e)
