"""Host-side profile (cProfile) of the README-sized dopri5 training step - a host-bound case: where does the Python time go?"""
import cProfile
import os
import pstats
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndcn_amd import graphs
from ndcn_amd.neural_dynamics import NDCN

dev = torch.device('cuda:0')
side, H, ticks = 20, 20, 80
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
n = side * side
torch.manual_seed(0)
model = NDCN(input_size=1, hidden_size=H, A=graphs.to_device(L, dev), num_classes=1, rtol=.01, atol=.001, method='dopri5').to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
x0 = torch.from_numpy(graphs.x0_blocks(side)[:n]).to(dev)
t = torch.linspace(0., 5., ticks).to(dev)
target = torch.rand(n, ticks, device=dev)


def step():
    opt.zero_grad()
    pred = model(t, x0).squeeze().t()
    loss = F.l1_loss(pred, target)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
