#!/usr/bin/env python3
"""The reference's one published workload, timed (README.md:64-73): dgnn.py --model differential_gcn on Cora, 100 epochs x 5 runs in
772 s = 1.54 s per epoch on the author's (unnamed) machine - and its Pubmed-topology sibling at the README width (BASELINE config 5).

One epoch = what dgnn.py:192-222 does: train forward + cross-entropy + backward through the dopri5 solve + Adam step, then an eval
forward.  Prints one JSON line per case: seconds per epoch on the HIP path (median over the timed epochs, train / eval split) and for
the CPU oracle (torch autograd through the restated solver, same model, same box, a few epochs), plus a determinism record: the same
seed twice -> are loss values, gradients after step 1 and the final parameters bit-identical?

    python tools/bench_dgnn.py [--case cora|pubmed_topology_H256|all] [--epochs 30] [--oracle-epochs 2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def load_case(name):
    """-> (csr arrays, features [n, d] float32, labels, idx_train, idx_val, n_classes) on the host"""
    from conftest import load_golden
    import scipy.sparse as sp
    if name == 'cora':
        d, g = load_golden('dataset_cora'), load_golden('operators_cora')
        n = int(g['n'])
        feats = sp.csr_matrix((d['feat_data'], d['feat_indices'].astype(np.int64), d['feat_indptr']), shape=tuple(d['feat_shape'])).toarray()
        op = (g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n))
        return op, feats.astype(np.float32), d['labels'].astype(np.int64), d['idx_train'].astype(np.int64), d['idx_val'].astype(np.int64), 7
    g = load_golden('operators_pubmed')
    n = int(g['n'])
    key = 'alpha00' if 'alpha00_indptr' in g else 'op'
    op = (g[key + '_indptr'], g[key + '_indices'], g[key + '_data'], (n, n))
    rng = np.random.default_rng(0)                                  # the feature blob is missing from the reference mount: synthetic
    feats = (rng.random((n, 500)) < 0.1).astype(np.float32) * rng.random((n, 500)).astype(np.float32)
    feats /= np.maximum(feats.sum(1, keepdims=True), 1e-6)          # row-normalised bag of words, 500 columns, 3 classes (Pubmed's shape)
    labels = rng.integers(0, 3, n)
    perm = rng.permutation(n)
    return op, feats, labels.astype(np.int64), perm[:60], perm[60:560], 3


HID, T_END, TICKS, RTOL, ATOL, LR, WD = 256, 1.2, 16, 0.1, 0.1, 0.01, 0.024          # the README command
NO_CONTROL = os.environ.get('NDCN_DGNN_CONTROL', '0') != '1'                          # (NDCN_DGNN_CONTROL=1: with the Linear - the determinism probe of W / b)


def build_hip(case, dev, seed=0):
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEBlock2, ODEFunc
    op, feats, labels, itr, iva, ncls = case
    adj = CsrOperator.from_arrays(op[0], op[1], op[2], op[3], dev)
    torch.manual_seed(seed)
    t = torch.linspace(0, T_END, TICKS).float().to(dev)
    model = nn.Sequential(nn.Linear(feats.shape[1], HID), nn.Tanh(),
                          ODEBlock2(ODEFunc(HID, adj, dropout=0.0, no_control=NO_CONTROL), t, rtol=RTOL, atol=ATOL, method='dopri5', terminal=True),
                          nn.Linear(HID, ncls)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
    return model, opt, torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(itr).to(dev), torch.from_numpy(iva).to(dev)


def epoch_hip(model, opt, x, y, itr, iva):
    model.train()
    opt.zero_grad()
    out = model(x)
    loss = F.cross_entropy(out[itr], y[itr])
    loss.backward()
    grads = [p.grad.detach().clone() if p.grad is not None else None for p in model.parameters()]
    opt.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    model.eval()
    with torch.no_grad():
        out = model(x)
        lv = F.cross_entropy(out[iva], y[iva])
    lv_host = float(lv)
    return float(loss), lv_host, grads, t1


def time_hip(case, dev, epochs):
    model, opt, x, y, itr, iva = build_hip(case, dev)
    tr, ev = [], []
    for e in range(epochs + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, _, t1 = epoch_hip(model, opt, x, y, itr, iva)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if e >= 3:
            tr.append(t1 - t0)
            ev.append(t2 - t1)
    return float(np.median(tr)), float(np.median(ev))


def determinism(case, dev, epochs, deterministic_flag):
    """the same seed twice: run A records, per epoch, the tensors along the step - encoder output (torch Linear + tanh), ODE block
    output (the HIP dopri5 solve), logits, loss, the gradient arriving at the ODE block's output and leaving its input (the HIP
    backward through the solver), every parameter gradient; run B compares with torch.equal as it goes and names the FIRST tensor
    that differs."""
    torch.use_deterministic_algorithms(deterministic_flag, warn_only=True)
    names = ['0.weight', '0.bias', '2.odefunc.wt.weight', '2.odefunc.wt.bias', '3.weight', '3.bias']
    record, first = [], None
    for run in range(2):
        model, opt, x, y, itr, iva = build_hip(case, dev, seed=0)
        keep = {}
        model[2].register_forward_hook(lambda m, i, o: keep.update(h_in=i[0].detach().clone(), h_out=o.detach().clone()))
        model[2].register_full_backward_hook(lambda m, gi, go: keep.update(g_h_out=go[0].detach().clone(),
                                                                          g_h_in=None if gi[0] is None else gi[0].detach().clone()))
        for e in range(epochs):
            model.train()
            opt.zero_grad()
            out = model(x)
            loss = F.cross_entropy(out[itr], y[itr])
            loss.backward()
            row = [('h_in', keep['h_in']), ('h_out', keep['h_out']), ('logits', out.detach().clone()), ('loss', loss.detach().clone()),
                   ('g_h_out', keep['g_h_out']), ('g_h_in', keep.get('g_h_in'))]
            row += [('grad ' + n, None if p.grad is None else p.grad.detach().clone()) for n, p in zip(names, model.parameters())]
            opt.step()
            if run == 0:
                record.append(row)
            elif first is None:
                for (n, a), (_, b) in zip(record[e], row):
                    if a is not None and not torch.equal(a, b):
                        first = {'epoch': e, 'tensor': n, 'max_abs_diff': float((a - b).abs().max()), 'max_abs': float(a.abs().max()),
                                 'elements_differing': int((a != b).sum()), 'elements': a.numel()}
                        break
        final = [p.detach().clone() for p in model.parameters()]
        if run == 0:
            final_a = final
    torch.use_deterministic_algorithms(False)
    return {'use_deterministic_algorithms': deterministic_flag, 'epochs': epochs, 'first_difference': first,
            'final_params_identical': all(torch.equal(a, b) for a, b in zip(final_a, final))}


def time_oracle(case, epochs):
    from oracle import ndcn_oracle as orc
    op, feats, labels, itr, iva, ncls = case
    torch.manual_seed(0)
    A = orc.coo_from_csr(op[0], op[1], op[2], op[3])
    l0, l3 = nn.Linear(feats.shape[1], HID), nn.Linear(HID, ncls)
    W, b = nn.Parameter(torch.zeros(HID, HID)), nn.Parameter(torch.zeros(HID))      # `wt` exists even with no_control (neural_dynamics.py:16)
    params = list(l0.parameters()) + [W, b] + list(l3.parameters())
    opt = torch.optim.Adam(params, lr=LR, weight_decay=WD)
    x, y = torch.from_numpy(feats), torch.from_numpy(labels)
    itr_t, iva_t = torch.from_numpy(itr), torch.from_numpy(iva)
    t = torch.linspace(0, T_END, TICKS)
    f = orc.OracleODEFunc(A, W, b, no_control=True)

    def fwd():
        h = torch.tanh(l0(x))
        return l3(orc.odeint(f, h, t, rtol=RTOL, atol=ATOL, method='dopri5')[-1])

    times = []
    for e in range(epochs + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = F.cross_entropy(fwd()[itr_t], y[itr_t])
        loss.backward()
        opt.step()
        with torch.no_grad():
            F.cross_entropy(fwd()[iva_t], y[iva_t])
        if e:
            times.append(time.perf_counter() - t0)
    return float(np.median(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', default='all')
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--oracle-epochs', type=int, default=2)
    ap.add_argument('--det-epochs', type=int, default=60)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for name in (['cora', 'pubmed_topology_H256'] if a.case == 'all' else [a.case]):
        case = load_case(name)
        tr, ev = time_hip(case, dev, a.epochs)
        rec = {'case': 'dgnn differential_gcn, ' + name, 'nodes': int(case[0][3][0]), 'hidden': HID, 'ticks': TICKS, 'method': 'dopri5 rtol=atol=0.1 no_control',
               'hip_s_per_epoch': round(tr + ev, 5), 'hip_train_s': round(tr, 5), 'hip_eval_s': round(ev, 5), 'timed_epochs': a.epochs}
        if name == 'cora':
            rec['readme_s_per_epoch'] = 1.544                       # README.md:72: 772 s for 5 x 100 epochs, hardware not named
        if a.oracle_epochs:
            rec['cpu_oracle_s_per_epoch'] = round(time_oracle(case, a.oracle_epochs), 3)
            rec['cpu_threads'] = torch.get_num_threads()
        rec['determinism'] = [determinism(case, dev, a.det_epochs, False), determinism(case, dev, a.det_epochs, True)]
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
