"""Adam step time of the heat driver's loop with the fixed-grid methods at the README size (400 x 20, 80 ticks) and at 100k x 256 (10 ticks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from ndcn_amd import graphs
from ndcn_amd.neural_dynamics import NDCN
dev = torch.device('cuda:0')
for side, H, ticks in ((20, 20, 80), (316, 256, 10)):
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side)); n = side * side
    A = graphs.to_device(L, dev)
    for method in ('euler', 'midpoint', 'rk4', 'dopri5'):
        torch.manual_seed(0)
        model = NDCN(input_size=1, hidden_size=H, A=A, num_classes=1, rtol=.01, atol=.001, method=method).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
        x0 = torch.from_numpy(graphs.x0_blocks(side)[:n]).to(dev); t = torch.linspace(0., 5., ticks).to(dev); target = torch.rand(n, ticks, device=dev)
        def step():
            opt.zero_grad(); F.l1_loss(model(t, x0).squeeze().t(), target).backward(); opt.step()
        for _ in range(3): step()
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print('%6d x %3d  %-8s %7.2f ms per Adam step' % (n, H, method, 1e3 * float(np.median(ts))), flush=True)
