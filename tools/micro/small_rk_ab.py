"""One-launch midpoint / RK4 reverse sweeps (solve_small.hip) against the library's multi-launch loops, by state size: Adam step of a 40-tick solve."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ndcn_amd import graphs
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
for side, H in ((8, 16), (10, 16), (12, 20), (14, 20), (17, 20), (20, 20)):
    op = graphs.to_device(graphs.normalized_laplacian(graphs.grid_8_neighbor(side)), dev)
    for method in ('midpoint', 'rk4'):
        row = []
        for one in ('1', '0'):
            os.environ['NDCN_SOLVE_SMALL_RK_GRAD'] = one
            os.environ['NDCN_SOLVE_SMALL_RK_MAX'] = '1000000'
            torch.manual_seed(0)
            f = ODEFunc(H, op).to(dev)
            x0 = torch.rand(side * side, H, device=dev, requires_grad=True)
            t = torch.linspace(0, 2, 41, device=dev)
            def step():
                f.zero_grad(); y = ode.odeint(f, x0, t, method=method); y.sum().backward()
            for _ in range(3): step()
            torch.cuda.synchronize(); ts = []
            for _ in range(10):
                t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row.append(1e3 * float(np.median(ts)))
        print('%4d x %2d (%5d elements) %-8s one launch %6.2f ms   loops %6.2f ms' % (side * side, H, side * side * H, method, row[0], row[1]), flush=True)
