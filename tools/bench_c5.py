#!/usr/bin/env python3
"""Config C5 timing (parity case, not the judged bench line): dgnn.py's differential_gcn hot path - ODEBlock2(ODEFunc
(no_control), terminal) on the Pubmed topology (19 717 nodes, operator from the committed fixture), H = 256, dopri5
rtol = atol = 0.1, t = linspace(0, 1.2, 16) (dgnn.py:173-182, README command) - eager launches vs hipGraph replay."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    from ndcn_amd.torchdiffeq._impl.odeint import DeviceSolver
    dev = torch.device('cuda:0')
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'operators_pubmed.npz')))
    n = int(g['n'])
    A = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    torch.manual_seed(0)
    f = ODEFunc(256, A, no_control=True).to(dev).eval()
    x0 = torch.rand(n, 256, device=dev)
    t = torch.linspace(0., 1.2, 16).tolist()
    out = torch.empty_like(x0)
    res = {'workload': 'C5: Pubmed topology %d nodes nnz %d, H=256, no_control, dopri5 rtol=atol=0.1, 16 ticks on [0,1.2]' % (n, A.nnz)}
    for name, use_graph in (('eager', False), ('hipgraph', True)):
        s = DeviceSolver(f, n, 'dopri5', .1, .1, use_graph=use_graph)
        times = []
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.begin(x0, t[0])
            for ti in t[1:]:
                s.advance(ti, out)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        st = s.stats()
        med = float(np.median(times[2:]))
        res[name] = {'ms_per_solve': round(1e3 * med, 4), 'steps': int(st['steps']), 'nfe': int(st['nfe']),
                     'node_states_per_s': round(n * st['steps'] / med, 1), 'us_per_step': round(1e6 * med / st['steps'], 2)}
        s.close()
    res['speedup'] = round(res['eager']['ms_per_solve'] / res['hipgraph']['ms_per_solve'], 3)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
