"""Counterpart of the reference's dgnn.py for `--model differential_gcn` (dgnn.py:159-182): node classification
with Linear+Tanh -> ODEBlock2(ODEFunc) terminal -> Linear, trained by backprop through the solver.

Kept: the flags of dgnn.py:24-70 that the differential model reads, cross-entropy on the training nodes, Adam
(lr, weight decay), the per-epoch log line and the test report (dgnn.py:192-237), the README command
(README.md:64), the `--dump` results file (dgnn.py:240-244,259-261); `--model resGCN` (dgnn.py:129-140) stacks ndcn_amd.ode_gcn.ResBlock.  The other
--model choices (GCN / DeepGCN* / odeGCN) are static baselines outside
the accelerated path.

    python -m ndcn_amd.drivers.dgnn --dataset cora --model differential_gcn --iter 5 --dropout 0 --hidden 256 \\
        --T 1.2 --time_tick 16 --epochs 100 --weight_decay 0.024 --no_control --method dopri5 --alpha 0 --data_dir data
"""
import argparse
import datetime
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from ..neural_dynamics import ODEBlock2, ODEFunc


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--no-cuda', action='store_true', default=False)
    p.add_argument('--fastmode', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=-1, help='Random seed.')
    p.add_argument('--epochs', type=int, default=200)
    p.add_argument('--rtol', type=float, default=0.1)
    p.add_argument('--atol', type=float, default=0.1)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--weight_decay', type=float, default=5e-4)
    p.add_argument('--hidden', type=int, default=16)
    p.add_argument('--dropout', type=float, default=0.5)
    p.add_argument('--dataset', type=str, default='cora')
    p.add_argument('--model', type=str, default='differential_gcn')
    p.add_argument('--iter', type=int, default=1, help='Number of experiments to conduct')
    p.add_argument('--dump', action='store_true', default=False)
    p.add_argument('--delta', type=float, default=1.0, help='Scale of signals from neighborhoods (parsed and dumped like the reference; its code never reads it)')
    p.add_argument('--sms', action='store_true', default=False, help='print the summary block once more at the end (the reference texts it; dgnn.py:286-289)')
    p.add_argument('--T', type=float, default=2., help='Terminal Time')
    p.add_argument('--time_tick', type=int, default=5)
    p.add_argument('--no_control', action='store_true', help='No control in DYnamics')
    p.add_argument('--method', type=str, default='dopri5', choices=['dopri5', 'euler', 'midpoint', 'rk4'])
    p.add_argument('--alpha', type=float, default=0.5, help='Tuning Matrix Operator')
    p.add_argument('--data_dir', type=str, default='data')
    p.add_argument('-nhl', '--nHiddenLayers', type=int, default=0, help='Number of Hidden layers.')
    p.add_argument('--normalize', action='store_true', default=False, help='Row normalize the feature in residual block')
    p.add_argument('--Euler', action='store_true', default=False, help='Euler step in forward method')
    return p


def accuracy(output, labels):
    return (output.max(1)[1] == labels).double().mean()          # utils.py:321-325


def main(argv=None, data=None, quiet=False):
    args = build_parser().parse_args(argv)
    if args.model not in ('differential_gcn', 'resGCN'):
        # GCN / DeepGCN* are static baselines outside SURVEY 8; `odeGCN` and `GCN` cannot run in the reference's own
        # dgnn.py either (its train() calls model(features) without the operator / time vector they need)
        raise NotImplementedError('--model differential_gcn and resGCN run on the accelerated path')
    assert torch.cuda.is_available() and not args.no_cuda, 'ndcn_amd runs on a ROCm device; there is no CPU path'
    device = torch.device('cuda:0')
    if args.seed >= 0:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    if data is None:
        from ..planetoid import load_data
        data = load_data(args.dataset, args.alpha, args.data_dir, device)
    adj, features, labels, idx_train, idx_val, idx_test = data
    say = (lambda *a, **k: None) if quiet else print

    # model and optimiser are built ONCE, ahead of the --iter loop, as in the reference (dgnn.py:128-183): later
    # iterations keep training the same model
    num_classes = int(labels.max().item()) + 1
    say('T : {}, time tick: {}'.format(args.T, args.time_tick))
    t = torch.linspace(0, args.T, args.time_tick).float().to(device)
    if args.model == 'resGCN':                                    # dgnn.py:129-140
        from ..ode_gcn import ResBlock
        model = nn.Sequential(
            nn.Linear(features.shape[1], args.hidden, bias=True), nn.ReLU(inplace=True),
            *[ResBlock(args.hidden, adj, dropout=args.dropout, normalize=args.normalize, Euler=args.Euler)
              for _ in range(args.nHiddenLayers)],
            nn.Linear(args.hidden, num_classes, bias=True)).to(device)
    else:
        model = nn.Sequential(
            nn.Linear(features.shape[1], args.hidden, bias=True), nn.Tanh(),
            ODEBlock2(ODEFunc(args.hidden, adj, dropout=args.dropout, no_control=args.no_control), t,
                      rtol=args.rtol, atol=args.atol, method=args.method, terminal=True),
            nn.Linear(args.hidden, num_classes, bias=True)).to(device)
    optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    fout = None
    if args.dump:                                                 # dgnn.py:240-244: one tab-separated line per iteration
        os.makedirs('results', exist_ok=True)
        fname = 'results/results_{}.txt'.format(datetime.datetime.now().__str__().replace(':', '-'))
        fout = open(fname, 'w')
        fout.write(vars(args).__str__() + '\n')
        fout.write('Time\tLoss\tAccuracy\tStep\n')

    accs, t0 = [], time.time()
    for it in range(args.iter):
        t_start = time.time()
        for epoch in range(args.epochs):
            te = time.time()
            model.train()
            optimizer.zero_grad()
            output = model(features)
            loss_train = F.cross_entropy(output[idx_train], labels[idx_train])
            acc_train = accuracy(output[idx_train], labels[idx_train])
            loss_train.backward()
            optimizer.step()
            if not args.fastmode:
                model.eval()
                with torch.no_grad():
                    output = model(features)
            loss_val = F.cross_entropy(output[idx_val], labels[idx_val])
            acc_val = accuracy(output[idx_val], labels[idx_val])
            say('ITER: {:04d}'.format(it + 1), 'Epoch: {:04d}'.format(epoch + 1),
                'loss_train: {:.4f}'.format(loss_train.item()), 'acc_train: {:.4f}'.format(acc_train.item()),
                'loss_val: {:.4f}'.format(loss_val.item()), 'acc_val: {:.4f}'.format(acc_val.item()),
                'time: {:.4f}s'.format(time.time() - te))
        say('Optimization Finished!')
        t_total = time.time() - t_start
        say('Total time elapsed: {:.4f}s'.format(t_total))
        model.eval()
        with torch.no_grad():
            output = model(features)
            loss_test = F.cross_entropy(output[idx_test], labels[idx_test])
            acc_test = accuracy(output[idx_test], labels[idx_test])
        print('Test set results:', 'loss= {:.4f}'.format(loss_test.item()), 'accuracy= {:.4f}'.format(acc_test.item()))
        accs.append(acc_test.item())
        if fout is not None:                                      # dgnn.py:259-261 (its time_step column is the constant 0)
            fout.write('{:.5f}\t{:.5f}\t{:.5f}\t{:.5f}\n'.format(t_total, loss_test.item(), acc_test.item(), 0))
            fout.flush()
    if fout is not None:
        fout.close()
        say('Dump results as: ' + fname)
    accs = np.array(accs)
    print('Total time: {:.4f}s;'.format(time.time() - t0))
    print('results: {:.3f}% (mean) +/- {:.3f}% (std), {:.3f}% (median);'.format(100 * accs.mean(), 100 * accs.std(), 100 * np.median(accs)))
    print('Min_Acc: {:.3f}%, Max_Acc: {:.3f}%'.format(100 * accs.min(), 100 * accs.max()))
    return accs


if __name__ == '__main__':
    main()
