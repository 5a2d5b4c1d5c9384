"""Gradients of a dopri5 solve (rejected steps, several ticks per step) from the three forms of the grad path - carry + multi-tick
dense (default), carry with single-tick dense, fan-out - each against torch autograd through the CPU oracle's restated solver."""
import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from conftest import load_golden
from oracle import ndcn_oracle as orc
from ndcn_amd import CsrOperator
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0'); T = torch.from_numpy
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
d = load_golden('fixed_rk4_equal')
for tt, rtol, atol in (([0., 0.01, 0.02, 0.9, 1.0, 2.5], 1e-5, 1e-7), ([0., 0.3, 0.6, 0.9, 1.0], 1e-3, 1e-5)):
    t = torch.tensor(tt)
    w = torch.randn(len(tt), *d['x0'].shape, generator=torch.Generator().manual_seed(3))
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], tuple(d['shape']))
    Wo, bo, xo = T(d['W']).clone().requires_grad_(True), T(d['b']).clone().requires_grad_(True), T(d['x0']).clone().requires_grad_(True)
    yo = orc.odeint(orc.OracleODEFunc(A, Wo, bo), xo, t, rtol=rtol, atol=atol, method='dopri5')
    (yo * w).sum().backward()
    ref = [xo.grad, Wo.grad, bo.grad]
    for name, env in (('carry + multi-tick', {}), ('carry, single-tick', {'NDCN_GRAD_MULTI_TICK': '0'}), ('fan-out', {'NDCN_GRAD_CARRY': '0'})):
        os.environ.update(env)
        f = ODEFunc(20, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)).to(dev)
        f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
        x0 = T(d['x0']).to(dev).requires_grad_(True)
        log = []
        y = ode.odeint(f, x0, t.to(dev), rtol=rtol, atol=atol, method='dopri5', step_log=log)
        (y * w.to(dev)).sum().backward()
        got = [x0.grad.cpu(), f.wt.weight.grad.cpu(), f.wt.bias.grad.cpu()]
        for k in env: del os.environ[k]
        print('rtol %g  %-20s attempts %2d  traj vs oracle %.2e  grad rel vs oracle %s' % (rtol, name, len(log) - 1, float((y.detach().cpu() - yo.detach()).abs().max()), ['%.2e' % rel(a, b) for a, b in zip(got, ref)]))
