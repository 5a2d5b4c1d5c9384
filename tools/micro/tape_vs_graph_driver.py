"""Gradient of the heat driver's first training steps (irregular ticks, NDCN encoder / decoder around the solve): native tape vs the
per-operation graph, at the initial parameters and after 60 Adam steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from ndcn_amd import graphs
from ndcn_amd.neural_dynamics import NDCN
dev = torch.device('cuda:0')
S, H = 20, 20
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(S)); n = S * S
A = graphs.to_device(L, dev)
rng = np.random.default_rng(0)
tt = np.linspace(0., 5., 120)
idx = np.sort(np.concatenate([[0], rng.choice(np.arange(1, 100), 79, replace=False)]))
t = torch.from_numpy(tt[idx].astype(np.float32)).to(dev)
x0 = torch.from_numpy(graphs.x0_blocks(S)[:n]).to(dev)
target = torch.rand(n, len(idx), device=dev)
torch.manual_seed(0)
model = NDCN(input_size=1, hidden_size=H, A=A, num_classes=1, rtol=.01, atol=.001, method='dopri5').to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
def grads(tape):
    os.environ['NDCN_GRAD_TAPE'] = tape
    model.zero_grad()
    loss = F.l1_loss(model(t, x0).squeeze().t(), target)
    loss.backward()
    return float(loss), [p.grad.detach().clone() for p in model.parameters()]
for stage in ('initial parameters', 'after 60 Adam steps'):
    la, ga = grads('1'); lb, gb = grads('0')
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(ga, gb))
    print('%-22s loss %.7f / %.7f   worst gradient difference %.2e of its tensor\'s maximum' % (stage, la, lb, worst), flush=True)
    if stage.startswith('initial'):
        os.environ['NDCN_GRAD_TAPE'] = '1'
        for _ in range(60):
            opt.zero_grad(); F.l1_loss(model(t, x0).squeeze().t(), target).backward(); opt.step()
