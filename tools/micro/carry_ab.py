import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
from conftest import load_golden
from ndcn_amd import CsrOperator
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0')
T = torch.from_numpy
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for name, scale, tt, rtol, atol in (('fixed_rk4_equal', 1.0, [0., 0.01, 0.02, 0.9, 1.0, 2.5], 1e-5, 1e-7), ('fixed_rk4_equal', 3.0, [0., 0.01, 0.02, 0.9, 1.0, 2.5], 1e-5, 1e-7),
                                   ('dopri5_tight', 1.0, None, None, None), ('fixed_rk4_equal', 1.0, [0., 0.5, 1.0], 1e-2, 1e-3)):
    d = load_golden(name)
    res = {}
    for flag in ('1', '0'):
        os.environ['NDCN_GRAD_CARRY'] = flag
        f = ODEFunc(20, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)).to(dev)
        f.load_state_dict({'wt.weight': T(d['W']) * scale, 'wt.bias': T(d['b'])})
        x0 = T(d['x0']).to(dev).requires_grad_(True)
        t = torch.tensor(tt, device=dev) if tt else T(d['t']).to(dev)
        log = []
        y = ode.odeint(f, x0, t, rtol=rtol or float(d['rtol']), atol=atol or float(d['atol']), method='dopri5', step_log=log)
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        (y * w).sum().backward()
        res[flag] = (y.detach().cpu(), log, [v.cpu().clone() for v in (x0.grad, f.wt.weight.grad, f.wt.bias.grad)])
    rej = sum(1 for r in res['1'][1] if r[0] != 'nfe' and r[2] == 0.0)
    print(name, scale, 'attempts', len(res['1'][1]) - 1, 'rejected', rej, 'traj equal', torch.equal(res['1'][0], res['0'][0]), 'rel', [rel(a, b) for a, b in zip(res['1'][2], res['0'][2])],
          'gmax', [float(b.abs().max()) for b in res['0'][2]])
