"""Sanity sweep (measurement aid): every public panel op at a large size, time and achieved GB/s against the bytes it must
move - outliers are the next thing to look at.  python tools/micro/op_sweep.py [side]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from ndcn_amd import hip, graphs
dev = torch.device('cuda:0')
side = int(sys.argv[1]) if len(sys.argv) > 1 else 700
n = side * side
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
A = graphs.to_device(L, dev)


def t(f, reps=10):
    for _ in range(2): f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps


def line(name, ms, byt):
    print('%-44s %8.3f ms  %7.0f GB/s' % (name, ms, byt / ms / 1e6))


for H in (1, 20, 64, 128, 256, 512):
    X = torch.rand(n, H, device=dev)
    line('spmm H=%d' % H, t(lambda: hip.spmm(A, X)), 8 * L.nnz + 8 * n * H)
for Hi, Ho in ((1, 20), (20, 20), (20, 1), (64, 64), (128, 128), (256, 256), (1, 256), (256, 1), (512, 512)):
    S = torch.rand(n, Hi, device=dev); W = torch.rand(Ho, Hi, device=dev); b = torch.rand(Ho, device=dev)
    line('linear %d->%d' % (Hi, Ho), t(lambda: hip.linear(S, W, b, relu=True)), 4 * n * (Hi + Ho))
H = 256
P = 4 * n * H
y0, y1 = torch.rand(n, H, device=dev), torch.rand(n, H, device=dev)
ks = [torch.rand(n, H, device=dev) for _ in range(7)]
cs = [np.float32(0.1 * (i + 1)) for i in range(7)]
line('combine 3 terms', t(lambda: hip.combine(y0, ks[:3], cs[:3])), 5 * P)
line('error 6 terms', t(lambda: hip.error(y0, y1, ks[:6], cs[:6], 1e-2, 1e-3)), 8 * P)
line('scaled_sumsq (a - b)', t(lambda: hip.scaled_sumsq(ks[0], ks[1], y0, 1e-2, 1e-3)), 3 * P)
fit = hip.interp_fit(y0, y1, ks, cs, 0.1)
line('interp_fit', t(lambda: hip.interp_fit(y0, y1, ks, cs, 0.1)), 12 * P)
xp = tuple(np.float32(v) for v in (0.1, 0.2, 0.3, 0.5, 1.0))
line('interp_eval', t(lambda: hip.interp_eval(fit, y0, xp)), 6 * P)
line('interp_direct', t(lambda: hip.interp_direct(y0, y1, ks, cs, 0.1, xp)), 10 * P)
for op, nk in ((0, 1), (5, 4)):
    line('fixed_stage op %d' % op, t(lambda: hip.fixed_stage(op, y0, *ks[:nk], dt=0.1)), (nk + 2) * P)
line('copy', t(lambda: hip.copy(y0)), 2 * P)
line('scale', t(lambda: hip.scale(y0, 0.3)), 2 * P)
line('relu_bwd', t(lambda: hip.relu_bwd(y0, y1)), 3 * P)
idx = torch.randint(0, n, (n // 4,), device=dev, dtype=torch.int32)
line('gather_rows n/4', t(lambda: hip.gather_rows(y0, idx)), 2 * (n // 4) * H * 4)
x1 = torch.rand(n, 1, device=dev)
line('gene_rhs (H=1)', t(lambda: hip.gene_rhs(A, x1)), 8 * L.nnz + 12 * n)
line('mutual_rhs (H=1)', t(lambda: hip.mutual_rhs(A, x1)), 8 * L.nnz + 12 * n)
from ndcn_amd.ode_gcn import row_normalization
line('row_normalization', t(lambda: row_normalization(y0)), 2 * P)
