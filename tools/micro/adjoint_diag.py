"""Where do odeint_adjoint and backprop-through-solver differ at a tight tolerance?  Per-gradient rel differences + a central
finite difference of the loss along a random direction of x0 and of W."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ndcn_amd import graphs
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
from ndcn_amd.torchdiffeq._impl import adjoint_fused
dev = torch.device('cuda:0')
H = 256
L = graphs.normalized_laplacian(graphs.grid_8_neighbor(24))
N = L.shape[0]
torch.manual_seed(5)
f = ODEFunc(H, graphs.to_device(L, dev)).to(dev)
x_init = torch.rand(N, H, generator=torch.Generator().manual_seed(6)).to(dev)
t = torch.tensor([0., 0.35, 0.8], device=dev)
wgt = torch.randn(3, N, H, generator=torch.Generator().manual_seed(7)).to(dev)
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))

def run(solver, fused, rtol, atol, x=None):
    adjoint_fused.ENABLED = fused
    for p in f.parameters():
        p.grad = None
    x0 = (x_init if x is None else x).clone().requires_grad_(True)
    y = solver(f, x0, t, rtol=rtol, atol=atol, method='dopri5')
    loss = (y * wgt).sum()
    loss.backward()
    return float(loss), [x0.grad.clone()] + [p.grad.clone() for p in f.parameters()]

for rtol, atol in ((1e-3, 1e-5), (1e-6, 1e-8)):
    la, ga = run(ode.odeint_adjoint, True, rtol, atol)
    lb, gb = run(ode.odeint_adjoint, False, rtol, atol)
    lc, gc = run(ode.odeint, True, rtol, atol)
    print('rtol', rtol, 'fused vs generic', [round(rel(a, b), 6) for a, b in zip(ga, gb)], 'fused vs backprop', [round(rel(a, c), 6) for a, c in zip(ga, gc)])
    d = torch.randn(N, H, generator=torch.Generator().manual_seed(9)).to(dev)
    eps = 1e-2
    with torch.no_grad():
        lp = float((ode.odeint(f, x_init + eps * d, t, rtol=rtol, atol=atol, method='dopri5') * wgt).sum())
        lm = float((ode.odeint(f, x_init - eps * d, t, rtol=rtol, atol=atol, method='dopri5') * wgt).sum())
    fd = (lp - lm) / (2 * eps)
    print('   dL/dx0 . d : finite difference %.5f | adjoint fused %.5f | generic %.5f | backprop %.5f' % (fd, float((ga[0] * d).sum()), float((gb[0] * d).sum()), float((gc[0] * d).sum())))
    dW = torch.randn(H, H, generator=torch.Generator().manual_seed(10)).to(dev) / 16
    W0 = f.wt.weight.detach().clone()
    with torch.no_grad():
        f.wt.weight.copy_(W0 + eps * dW); lp = float((ode.odeint(f, x_init, t, rtol=rtol, atol=atol, method='dopri5') * wgt).sum())
        f.wt.weight.copy_(W0 - eps * dW); lm = float((ode.odeint(f, x_init, t, rtol=rtol, atol=atol, method='dopri5') * wgt).sum())
        f.wt.weight.copy_(W0)
    print('   dL/dW . dW : finite difference %.5f | adjoint fused %.5f | generic %.5f | backprop %.5f' % ((lp - lm) / (2 * eps), float((ga[1] * dW).sum()), float((gb[1] * dW).sum()), float((gc[1] * dW).sum())))
