"""Counterpart of the reference's three experiment drivers (heat_dynamics.py / gene_dynamics.py /
mutualistic_dynamics.py - they differ only in the ground-truth dynamics) on the HIP path.

Kept from the reference: flag names and defaults (heat_dynamics.py:19-64), the graph choices
(:83-110), the equal / irregular time split (:121-147), the operator choices (:150-167), the three-block
initial image (:178-182), the `--layout` node re-labelling of the non-grid networks (:90-109;
utils_in_learn_dynamics.py:212-247), the truth solve with dopri5 at odeint's default tolerances (:207-209), the model
variants (:245-268), Adam + L1 / relative-L1 (:295-321), the log line formats (:374-388), the dump dictionary
keys and file naming (:300-311, :434-438).  Different by construction: graphs and operators are built in O(nnz)
(ndcn_amd/graphs.py, so --n can be 10^6), everything runs on the ROCm device, `--seed` also seeds torch / numpy
(the reference leaves them unseeded), the RNN-GNN baselines and --viz are out of scope.

    python -m ndcn_amd.drivers.heat_dynamics --network grid --sampled_time equal --baseline ndcn --gpu 0
"""
import argparse
import datetime
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from .. import graphs
from .. import torchdiffeq as ode
from ..neural_dynamics import NDCN
from ..ops import hip

TITLES = {'heat': 'Heat Diffusion Dynamic Case', 'gene': 'Gene Regulatory Dynamic Case',
          'mutualistic': 'Mutualistic Interaction Dynamic Case'}


def build_parser(kind):
    p = argparse.ArgumentParser(TITLES[kind])
    p.add_argument('--method', type=str, default='euler',
                   choices=['dopri5', 'adams', 'explicit_adams', 'fixed_adams', 'tsit5', 'euler', 'midpoint', 'rk4'])
    p.add_argument('--rtol', type=float, default=0.01)
    p.add_argument('--atol', type=float, default=0.001)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--weight_decay', type=float, default=1e-3)
    p.add_argument('--dropout', type=float, default=0)
    p.add_argument('--hidden', type=int, default=20)
    p.add_argument('--time_tick', type=int, default=100)
    p.add_argument('--sampled_time', type=str, choices=['irregular', 'equal'], default='irregular')
    p.add_argument('--niters', type=int, default=2000)
    p.add_argument('--test_freq', type=int, default=20)
    p.add_argument('--viz', action='store_true')
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--adjoint', action='store_true')
    p.add_argument('--n', type=int, default=400, help='Number of nodes')
    p.add_argument('--sparse', action='store_true')
    p.add_argument('--network', type=str, choices=['grid', 'random', 'power_law', 'small_world', 'community'], default='grid')
    p.add_argument('--layout', type=str, choices=['community', 'degree'], default='community')
    p.add_argument('--seed', type=int, default=0, help='Random Seed')
    p.add_argument('--T', type=float, default=5., help='Terminal Time')
    p.add_argument('--operator', type=str, choices=['lap', 'norm_lap', 'kipf', 'norm_adj'], default='norm_lap')
    p.add_argument('--baseline', type=str, default='ndcn',
                   choices=['ndcn', 'no_embed', 'no_control', 'no_graph', 'lstm_gnn', 'rnn_gnn', 'gru_gnn'])
    p.add_argument('--dump', action='store_true', help='Save Results')
    return p


def time_split(args, rng):
    """heat_dynamics.py:121-147."""
    if args.sampled_time == 'equal':
        t = torch.linspace(0., args.T, args.time_tick)
        id_train = list(range(int(args.time_tick * 0.8)))
        id_test = list(range(int(args.time_tick * 0.8), args.time_tick))
        return t, id_train, id_test, None
    dense = torch.linspace(0., args.T, args.time_tick * 10).numpy()
    t = torch.tensor(np.sort(rng.permutation(dense)[:int(args.time_tick * 1.2)]))
    t[0] = 0
    id_test = list(range(args.time_tick, int(args.time_tick * 1.2)))
    id_test2 = sorted(rng.permutation(range(1, args.time_tick))[:int(args.time_tick * 0.2)].tolist())
    id_train = sorted(set(range(args.time_tick)) - set(id_test2))
    return t, id_train, id_test, id_test2


def truth_rhs(kind, A_op, L_op):
    """The three ground-truth right-hand sides as O(nnz) HIP kernels."""
    if kind == 'heat':
        return lambda t, x: hip.spmm(L_op, x, alpha=-1.0)             # heat_dynamics.py:189-204, k = 1
    if kind == 'gene':
        return lambda t, x: hip.gene_rhs(A_op, x, b=1.0, f=1.0, h=2.0)  # gene_dynamics.py:186-205
    return lambda t, x: hip.mutual_rhs(A_op, x)                       # mutualistic_dynamics.py:186-216


def main(kind, argv=None):
    args = build_parser(kind).parse_args(argv)
    if args.baseline in ('lstm_gnn', 'rnn_gnn', 'gru_gnn'):
        raise NotImplementedError('the discrete RNN-GNN baselines (TemporalGCN) are outside the accelerated path')
    assert torch.cuda.is_available() and args.gpu >= 0, 'ndcn_amd runs on a ROCm device (--gpu >= 0); there is no CPU path'
    device = torch.device('cuda:%d' % args.gpu)
    torch.manual_seed(args.seed)
    rng = np.random.RandomState(args.seed)

    # ---- graph, operators, initial value
    print('Choose graph: ' + args.network)
    # every network but the grid is re-labelled by --layout, as heat_dynamics.py:90,95,100,109 (the x0 image is indexed
    # by node position, so the node order is part of the learning problem)
    A = graphs.make_graph(args.network, args.n, seed=args.seed, layout=args.layout)
    n = A.shape[0]
    S = int(np.ceil(np.sqrt(n)))
    L = graphs.laplacian(A)
    names = {'lap': 'Laplacian', 'kipf': 'Kipf', 'norm_adj': 'Normalized Adjacency', 'norm_lap': 'Normalized Laplacian'}
    print('Graph Operator%s: %s' % ('[Default]' if args.operator == 'norm_lap' else '', names[args.operator]))
    OM = graphs.to_device(graphs.make_operator(A, args.operator), device)
    A_op, L_op = graphs.to_device(A, device), graphs.to_device(L, device)
    x0 = torch.from_numpy(graphs.x0_blocks(S)[:n]).to(device)

    # ---- time grid and ground truth (odeint defaults: rtol 1e-7, atol 1e-9)
    print('Build %s -time dynamics' % ('Equally-sampled' if args.sampled_time == 'equal' else 'irregularly-sampled'))
    t, id_train, id_test, id_test2 = time_split(args, rng)
    t = t.to(device)
    with torch.no_grad():
        solution_numerical = ode.odeint(truth_rhs(kind, A_op, L_op), x0, t, method='dopri5')
        print(solution_numerical.shape)
    true_y = solution_numerical.squeeze().t()
    true_y_train, true_y_test = true_y[:, id_train], true_y[:, id_test]
    true_y_test2 = true_y[:, id_test2] if id_test2 is not None else None
    t_train = t[id_train]

    # ---- model
    print('Choose model:' + args.baseline)
    hidden = 1 if args.baseline == 'no_embed' else args.hidden
    model = NDCN(input_size=1, hidden_size=hidden, A=OM, num_classes=1, dropout=args.dropout,
                 no_embed=args.baseline == 'no_embed', no_graph=args.baseline == 'no_graph',
                 no_control=args.baseline == 'no_control', rtol=args.rtol, atol=args.atol, method=args.method).to(device)
    num_paras = sum(p.numel() for p in model.parameters())
    print({'Total': num_paras, 'Trainable': sum(p.numel() for p in model.parameters() if p.requires_grad)})

    t_start = time.time()
    optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    criterion = F.l1_loss
    results = {'args': args.__dict__, 'v_iter': [], 'abs_error': [], 'rel_error': [], 'true_y': [true_y],
               'predict_y': [], 'abs_error2': [], 'rel_error2': [], 'predict_y2': [], 'model_state_dict': [],
               'total_time': []}

    def evaluate():
        with torch.no_grad():
            pred = model(t, x0).squeeze().t()
            loss = criterion(pred[:, id_test], true_y_test)
            rel = loss / true_y_test.mean()
            loss2 = rel2 = None
            if id_test2 is not None:
                loss2 = criterion(pred[:, id_test2], true_y_test2)
                rel2 = loss2 / true_y_test2.mean()
        return pred, loss, rel, loss2, rel2

    def report(itr, loss_train, rel_train, loss, rel, loss2, rel2):
        if id_test2 is not None:
            print('Iter {:04d}| Train Loss {:.6f}({:.6f} Relative) | Test Loss {:.6f}({:.6f} Relative) '
                  '| Test Loss2 {:.6f}({:.6f} Relative) | Time {:.4f}'
                  .format(itr, loss_train.item(), rel_train.item(), loss.item(), rel.item(), loss2.item(), rel2.item(),
                          time.time() - t_start), flush=True)
        else:
            print('Iter {:04d}| Train Loss {:.6f}({:.6f} Relative) | Test Loss {:.6f}({:.6f} Relative) | Time {:.4f}'
                  .format(itr, loss_train.item(), rel_train.item(), loss.item(), rel.item(), time.time() - t_start),
                  flush=True)

    itr, loss_train, rel_train = 0, torch.zeros(()), torch.zeros(())
    for itr in range(1, args.niters + 1):
        optimizer.zero_grad()
        pred_y = model(t_train, x0).squeeze().t()
        loss_train = criterion(pred_y, true_y_train)
        rel_train = loss_train.detach() / true_y_train.mean()
        loss_train.backward()
        optimizer.step()
        if itr % args.test_freq == 0:
            pred, loss, rel, loss2, rel2 = evaluate()
            if args.dump:
                results['v_iter'].append(itr)
                results['abs_error'].append(loss.item())
                results['rel_error'].append(rel.item())
                results['predict_y'].append(pred[:, id_test].cpu())
                results['model_state_dict'].append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
                if id_test2 is not None:
                    results['abs_error2'].append(loss2.item())
                    results['rel_error2'].append(rel2.item())
                    results['predict_y2'].append(pred[:, id_test2].cpu())
            report(itr, loss_train.detach(), rel_train, loss, rel, loss2, rel2)

    pred, loss, rel, loss2, rel2 = evaluate()
    report(itr, loss_train.detach(), rel_train, loss, rel, loss2, rel2)
    t_total = time.time() - t_start
    print('Total Time {:.4f}'.format(t_total))
    if args.dump:
        results['total_time'] = t_total
        results['true_y'] = [true_y.cpu()]
        results_dir = r'results/%s/' % kind + args.network
        os.makedirs(results_dir, exist_ok=True)
        path = results_dir + r'/result_' + datetime.datetime.now().strftime('%m%d-%H%M%S') + '.' + args.baseline
        torch.save(results, path)
        print('Dump results as: ' + path)
    return {'loss': loss.item(), 'rel': rel.item(), 'train_loss': float(loss_train), 'params': num_paras}
