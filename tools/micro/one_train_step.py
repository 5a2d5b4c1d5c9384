"""N Adam steps of the 100k-node dopri5 training case (for rocprofv3 traces)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from ndcn_amd import graphs
from ndcn_amd.neural_dynamics import NDCN
side, ticks, dev = int(os.environ.get('SIDE', 316)), 10, torch.device('cuda:0')
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
n = side * side
A = graphs.to_device(L, dev)
torch.manual_seed(0)
model = NDCN(input_size=1, hidden_size=256, A=A, num_classes=1, rtol=.01, atol=.001, method='dopri5').to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
x0 = torch.from_numpy(graphs.x0_blocks(side)[:n]).to(dev)
t = torch.linspace(0., 5., ticks).to(dev)
target = torch.rand(n, ticks, device=dev)
import time
for i in range(int(os.environ.get('STEPS', 8))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = F.l1_loss(model(t, x0).squeeze(-1).t(), target)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print('step %d %.3f ms' % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
