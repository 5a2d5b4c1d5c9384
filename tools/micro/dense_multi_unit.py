import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
import numpy as np, torch
from ndcn_amd import hip
dev = torch.device('cuda:0')
f32 = np.float32
g = torch.Generator().manual_seed(0)
n = 4096
y0, y1 = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
ks = [torch.randn(n, generator=g).to(dev) for _ in range(7)]
gs = [torch.randn(n, generator=g).to(dev) for _ in range(3)]
xs = [f32(0.2), f32(0.55), f32(0.9)]
dt = f32(0.37)
acc = [torch.randn(n, generator=g).to(dev) for _ in range(7)]
a0, a1 = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
gy0, gy1, gk, dxs, ddt = hip.interp_bwd_multi(gs, y0, y1, ks, dt, xs, True, True, [True] * 7, accs=acc, acc_y0=a0, acc_y1=a1)
ry0, ry1, rk, rdx, rdt = a0.clone(), a1.clone(), [a.clone() for a in acc], [], 0.0
for gt, x in zip(gs, xs):
    b0, b1, bk, d_x, d_dt = hip.interp_bwd(gt, y0, y1, ks, dt, x, True, True, [True] * 7)
    ry0 += b0; ry1 += b1
    for j in range(7): rk[j] += bk[j]
    rdx.append(d_x); rdt += d_dt
print('gy0', float((gy0 - ry0).abs().max()), 'gy1', float((gy1 - ry1).abs().max()), 'gk', [float((a - b).abs().max()) for a, b in zip(gk, rk)])
print('dx', dxs, rdx, 'ddt', ddt, rdt)
