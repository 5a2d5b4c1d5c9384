"""README-sized dopri5 solves: the NDCN ODEFunc (400 x 20, 80 ticks, rtol .01) through the device-resident solver, and the
heat ground-truth solve of the drivers (400 x 1, 100 ticks, rtol 1e-7 / atol 1e-9: heat_dynamics.py:207-209) through the generic path."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ndcn_amd import graphs, hip
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
G = graphs.grid_8_neighbor(20)
L = graphs.normalized_laplacian(G)
torch.manual_seed(0)
f = ODEFunc(20, graphs.to_device(L, dev)).to(dev).eval()
x0 = torch.rand(400, 20, device=dev)
t = torch.linspace(0., 5., 81).to(dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    log = []
    ode.odeint(f, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=log)
    print('NDCN ODEFunc dopri5 80 ticks: %.3f ms per solve (%d attempts)' % (timeit(lambda: ode.odeint(f, x0, t, rtol=.01, atol=.001, method='dopri5')), len(log) - 1))
    Lap = graphs.to_device(graphs.laplacian(G), dev)
    xt = torch.from_numpy(graphs.x0_blocks(20)[:400]).to(dev).view(-1, 1)
    tt = torch.linspace(0., 5., 100).to(dev)
    log = []
    ode.odeint(lambda s, x: hip.spmm(Lap, x, alpha=-1.0), xt, tt, method='dopri5', step_log=log)
    print('heat truth solve (400 x 1, rtol 1e-7): %.3f ms per solve (%d attempts)' % (timeit(lambda: ode.odeint(lambda s, x: hip.spmm(Lap, x, alpha=-1.0), xt, tt, method='dopri5'), 3), len(log) - 1))
