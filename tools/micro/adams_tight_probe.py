import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from conftest import load_golden
from ndcn_amd import CsrOperator
from ndcn_amd import torchdiffeq as ode
from ndcn_amd.neural_dynamics import ODEFunc
dev = torch.device('cuda:0'); T = torch.from_numpy
d = load_golden('adams_tight')
n = d['x0'].shape[0]
A = CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], (n, n), dev)
f = ODEFunc(d['x0'].shape[1], A, no_control=bool(d['no_control'])).to(dev).eval()
f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
log = []
with torch.no_grad():
    y = ode.odeint(f, T(d['x0']).to(dev), T(d['t']).to(dev), rtol=float(d['rtol']), atol=float(d['atol']), method='adams', step_log=log)
nfe = dict([log.pop()])['nfe']
ref, got = d['steplog'], np.array(log)
print('nfe', nfe, 'ref', int(d['nfe']), 'attempts', got.shape[0], 'ref', ref.shape[0])
m = min(len(ref), len(got))
diff = np.nonzero((got[:m, 2:4] != ref[:m, 2:4]).any(1))[0]
print('first differing attempt', diff[:3], 'of', m)
if diff.size:
    i = diff[0]
    for j in range(max(0, i - 2), min(m, i + 3)):
        print(j, 'got', got[j], 'ref', ref[j])
print('traj max abs diff', np.abs(y.cpu().numpy() - d['traj']).max())
