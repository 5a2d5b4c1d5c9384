import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndcn_amd import graphs, CsrOperator
from ndcn_amd.neural_dynamics import ODEFunc
from ndcn_amd.torchdiffeq import odeint
dev = torch.device('cuda:0')
for net, n in (('grid', 1000000), ('random', 100000)):
    L = graphs.normalized_laplacian(graphs.make_graph(net, n, seed=0))
    A = CsrOperator.from_scipy(L, dev)
    torch.manual_seed(0)
    f = ODEFunc(256, A).to(dev)
    x0 = torch.rand(L.shape[0], 256, device=dev)
    t = torch.linspace(0, 5, 26 if n > 500000 else 100).to(dev)
    with torch.no_grad():
        odeint(f, x0, t[:3], method='rk4'); torch.cuda.synchronize()
        t0 = time.perf_counter(); y = odeint(f, x0, t, method='rk4')[-1]; torch.cuda.synchronize(); dtw = time.perf_counter() - t0
    steps = t.numel() - 1
    print('%s n=%d rk4: %d steps %.2f ms/step  %.1f M node-states/s' % (net, L.shape[0], steps, 1e3 * dtw / steps, L.shape[0] * steps / dtw / 1e6), flush=True)
