#!/usr/bin/env python3
"""Training-step timing (measurement aid, not the judged bench line): what every reference driver does per iteration
(heat_dynamics.py:326-335): NDCN forward through the solver, L1 loss, backward through every solver op, Adam step.
Two cases: C1 (400-node grid, H = 20, Euler on linspace(0,5,80) - the README command) and a 100k-node grid with H = 256
(Euler, 10 ticks).  The CPU oracle (torch autograd through the restated solver) runs the same step where it fits."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import torch
import torch.nn.functional as F


def one_case(name, side, H, ticks, method, dev, cpu_reps):
    from ndcn_amd import graphs
    from ndcn_amd.neural_dynamics import NDCN
    from oracle import ndcn_oracle as orc
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(side))
    n = side * side
    A = graphs.to_device(L, dev)
    torch.manual_seed(0)
    model = NDCN(input_size=1, hidden_size=H, A=A, num_classes=1, rtol=.01, atol=.001, method=method).to(dev)
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
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    # every timed repetition is the SAME Adam step (parameters and optimizer moments put back before it, outside the timed region):
    # an adaptive solve's cost moves with the parameters - 5 or 6 attempted dopri5 steps from one training step to the next, 31 vs 36 ms
    # at 100k nodes - and a median over consecutive training steps then measures which kind was in the majority
    import copy
    frozen = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))

    def rewind():
        model.load_state_dict(frozen[0])
        opt.load_state_dict(copy.deepcopy(frozen[1]))
        torch.cuda.synchronize()

    times = []
    for _ in range(10):
        rewind()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    rewind()
    res = {'case': name, 'n': n, 'H': H, 'method': method, 'ticks': ticks, 'gpu_ms_per_adam_step': round(1e3 * float(np.median(times)), 3)}
    if n * H >= (1 << 20):
        # a roofline per kernel family of the step (HIP events around every library launch; algorithmic bytes of each launch over
        # its duration against the 8 TB/s HBM peak): the backward kernels - combine_bwd, error_bwd, dense_bwd (dense output),
        # linear_gs, linear_wgrad - next to the forward ones
        import _prof
        bd, tot = _prof.breakdown(step)
        res['library_kernel_ms_per_step'] = round(tot, 2)
        res['rhs_evaluations_per_step'] = sum(v['launches'] for k, v in bd.items() if k in ('rhs_fused', 'rhs_adjoint_forward_half'))
        res['breakdown'] = bd
        res['roofline'] = _prof.roofline_of(bd, tot)
    if cpu_reps:
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        Ac = orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape)
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        optc = torch.optim.Adam(list(sd.values()), lr=0.01, weight_decay=1e-3)
        xc, tc, tg = x0.cpu(), t.cpu(), target.cpu()
        ct = []
        for r in range(1 + cpu_reps):
            t0 = time.perf_counter()
            optc.zero_grad()
            pred = orc.ndcn_forward(sd, Ac, tc, xc, method).squeeze().t()
            F.l1_loss(pred, tg).backward()
            optc.step()
            if r:
                ct.append(time.perf_counter() - t0)
        res['cpu_oracle_ms_per_adam_step'] = round(1e3 * float(np.median(ct)), 2)
        res['cpu_threads'] = torch.get_num_threads()
    return res


def main():
    dev = torch.device('cuda:0')
    out = [one_case('C1 heat_dynamics --network grid --baseline ndcn (README)', 20, 20, 80, 'euler', dev, 3),
           one_case('C1 with dopri5', 20, 20, 80, 'dopri5', dev, 3),
           one_case('100k-node grid, H=256', 316, 256, 10, 'euler', dev, 1),
           one_case('100k-node grid, H=256, dopri5', 316, 256, 10, 'dopri5', dev, 1)]
    # A/B: the solver's panel-op VJPs as autograd through torch expressions instead of the csrc/rk_bwd.hip kernels
    os.environ['NDCN_VJP'] = 'torch'
    for r in (one_case('C1 with dopri5 [NDCN_VJP=torch]', 20, 20, 80, 'dopri5', dev, 0),
              one_case('100k-node grid, H=256, dopri5 [NDCN_VJP=torch]', 316, 256, 10, 'dopri5', dev, 0)):
        out.append(r)
    os.environ.pop('NDCN_VJP')
    for r in out:
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
