import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ndcn_amd import graphs
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(20))
torch.manual_seed(0)
f = ODEFunc(20, graphs.to_device(L, dev)).to(dev).eval()
x0 = torch.rand(400, 20, device=dev)
for method in ('euler', 'rk4'):
    t = torch.linspace(0., 5., 81).to(dev)
    with torch.no_grad():
        for _ in range(5): ode.odeint(f, x0, t, method=method)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20): ode.odeint(f, x0, t, method=method)
        e1.record(); torch.cuda.synchronize()
        print(method, 'forward 80 ticks: wall %.3f ms  gpu %.3f ms per solve' % ((time.perf_counter() - t0) / 20 * 1e3, e0.elapsed_time(e1) / 20))
