import os, sys, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tests')
from conftest import load_golden
from ndcn_amd import CsrOperator
from ndcn_amd.drivers import dgnn
dev = torch.device('cuda:0')
d = load_golden('dataset_cora'); g = load_golden('operators_cora'); n = int(g['n'])
adj = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
feats = sp.csr_matrix((d['feat_data'], d['feat_indices'].astype(np.int64), d['feat_indptr']), shape=tuple(d['feat_shape']))
data = (adj, torch.from_numpy(feats.toarray()).to(dev), torch.from_numpy(d['labels'].astype(np.int64)).to(dev),
        torch.from_numpy(d['idx_train'].astype(np.int64)).to(dev), torch.from_numpy(d['idx_val'].astype(np.int64)).to(dev),
        torch.from_numpy(d['idx_test'].astype(np.int64)).to(dev))
accs = dgnn.main(['--dataset', 'cora', '--model', 'differential_gcn', '--iter', '8', '--dropout', '0', '--hidden', '256',
                  '--T', '1.2', '--time_tick', '16', '--epochs', '100', '--weight_decay', '0.024', '--no_control',
                  '--method', 'dopri5', '--alpha', '0', '--seed', '0'], data=data, quiet=True)
print('FUSED_ERROR', os.environ.get('NDCN_GRAD_FUSED_ERROR', '1'), np.round(accs, 3), 'mean %.4f' % accs.mean())
