"""Where does the host time of the C1 dopri5 Adam step go?  (400 nodes x 20 hidden, 80 ticks: launch- and interpreter-bound)"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from ndcn_amd import graphs
from ndcn_amd.neural_dynamics import NDCN
dev = torch.device('cuda:0')
side, H, ticks = 20, 20, int(os.environ.get('TICKS', 80))
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
n = side * side
A = graphs.to_device(L, dev)
torch.manual_seed(0)
model = NDCN(input_size=1, hidden_size=H, A=A, num_classes=1, rtol=.01, atol=.001, method='dopri5').to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-3)
x0 = torch.from_numpy(graphs.x0_blocks(side)[:n]).to(dev)
t = torch.linspace(0., 5., ticks).to(dev)
target = torch.rand(n, ticks, device=dev)
def fwd():
    return F.l1_loss(model(t, x0).squeeze().t(), target)
for _ in range(5):
    opt.zero_grad(); fwd().backward(); opt.step()
torch.cuda.synchronize()
tf = tb = to = 0.0
R = 20
for _ in range(R):
    opt.zero_grad()
    t0 = time.perf_counter(); loss = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
log = []
from ndcn_amd import torchdiffeq as ode
print('forward %.2f ms  backward %.2f ms  adam %.2f ms' % (1e3 * tf / R, 1e3 * tb / R, 1e3 * to / R))
for name, fn in (('forward', lambda: fwd()), ('backward', None)):
    pr = cProfile.Profile()
    if name == 'forward':
        pr.enable(); [fn() for _ in range(10)]; torch.cuda.synchronize(); pr.disable()
    else:
        losses = [fwd() for _ in range(10)]
        pr.enable(); [l.backward() for l in losses]; torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    print('====', name, '(10 repetitions)'); print('\n'.join(s.getvalue().splitlines()[:40]))
