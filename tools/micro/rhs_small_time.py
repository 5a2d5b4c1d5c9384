"""Timing aid: the whole ODEFunc in one launch for narrow panels (rhs_small.hip), N = 10^5-node grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ndcn_amd import hip, graphs, _lib
dev = torch.device('cuda:0')
sides = [int(v) for v in sys.argv[1:]] or [316]
for side, H in [(sd, h) for sd in sides for h in (20, 64, 128, 256)]:
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    A = graphs.to_device(L, dev)
    n = side * side
    X = torch.rand(n, H, device=dev); W = torch.randn(H, H, device=dev) / H ** .5; b = torch.randn(H, device=dev)
    for _ in range(3): hip.rhs(A, X, W, b)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): hip.rhs(A, X, W, b)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 50
    byt = 8 * L.nnz + 8 * n * H
    print('n=%6d H=%3d  %.4f ms  %.0f GB/s algorithmic  path %d' % (n, H, ms, byt / ms / 1e6, _lib.load().ndcn_debug_last_rhs_path()))
