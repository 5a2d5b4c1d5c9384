"""One README-sized Euler training solve (400 nodes, H = 20, 80 ticks): forward + backward launches of solve_small.hip; with
NDCN_SS_DEBUG=1 the kernels print their cycle accounting."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ndcn_amd import graphs
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(20))
torch.manual_seed(0)
f = ODEFunc(20, graphs.to_device(L, dev)).to(dev)
x0 = torch.rand(400, 20, device=dev, requires_grad=True)
t = torch.linspace(0., 5., 81).to(dev)
w = torch.rand(81, 400, 20, device=dev)
for _ in range(3):
    f.zero_grad(); (ode.odeint(f, x0, t, method='euler') * w).sum().backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    f.zero_grad(); (ode.odeint(f, x0, t, method='euler') * w).sum().backward()
torch.cuda.synchronize()
print('euler 80 ticks forward + backward: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
