import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ndcn_amd import graphs
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
for side, H, n_ticks in ((12, 16, 150), (12, 16, 100), (12, 20, 150), (12, 16, 128), (12, 16, 129)):
    op = graphs.normalized_laplacian(graphs.grid_8_neighbor(side)); n = side * side
    ticks = torch.linspace(0., 3.0, n_ticks + 1)
    x0h = torch.rand(n, H, generator=torch.Generator().manual_seed(2))
    w = torch.randn(n_ticks + 1, n, H, generator=torch.Generator().manual_seed(1))
    A = graphs.to_device(op, dev)
    res = {}
    for keep in ('1', '0'):
        os.environ['NDCN_SOLVE_SMALL_KEEP'] = keep
        torch.manual_seed(0)
        f = ODEFunc(H, A).to(dev)
        x0 = x0h.clone().to(dev).requires_grad_(True)
        y = ode.odeint(f, x0, ticks.to(dev), method='euler')
        (y * w.to(dev)).sum().backward()
        res[keep] = (y.detach().cpu(), x0.grad.cpu(), f.wt.weight.grad.cpu(), f.wt.bias.grad.cpu())
    print(side, H, n_ticks, [float((a - b).abs().max()) for a, b in zip(res['1'], res['0'])], [float(b.abs().max()) for b in res['0']])
