import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from ndcn_amd import graphs, hip, _lib
dev = torch.device('cuda:0')
H = 256
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(48))
A = graphs.to_device(L, dev)
torch.manual_seed(0)
lin = torch.nn.Linear(H, H)
W0, b = lin.weight.detach().clone(), lin.bias.detach().clone()
X0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(5))
for ch in (0, 5, 100, 255):
    for p in (2, 4, 8, 12):
        for wsmall in (False, True):
            X = X0.clone(); X[:, ch] *= 2.0 ** p
            W = W0.clone()
            if wsmall: W[:, ch] *= 2.0 ** -p
            got = hip.rhs(A, X.to(dev), W.to(dev), b.to(dev))
            path = int(_lib.load().ndcn_debug_last_rhs_path())
            S = hip.spmm(A, X.to(dev)).double()
            ref = torch.relu(S @ W.to(dev).double().t() + b.to(dev).double())
            mag = S.abs() @ W.to(dev).double().abs().t() + b.to(dev).double().abs()
            r = ((got.double() - ref).abs() / mag).max().item()
            print('ch %3d x2^%-2d wsmall %d path %d  max err/mag %.2e' % (ch, p, wsmall, path, r), flush=True)
