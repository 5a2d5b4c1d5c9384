"""Planetoid citation datasets (Cora / Citeseer / Pubmed) as the reference's utils.load_data assembles them
(utils.py:91-230), restated for current scipy (the reference's own loader raises under scipy >= 1.13:
`sp.csr_matrix((ones, zip(*row_col)))`).  SURVEY 8f rank 3: needed by the dgnn counterpart.

    adj, features, labels, idx_train, idx_val, idx_test = load_data('cora', alpha=0.0, data_dir='data')

Returns the propagation operator as a ndcn_amd CsrOperator (the reference returns a torch sparse COO tensor),
row-normalised dense features, integer labels and the three index tensors.
"""
import os
import pickle
import sys

import numpy as np
import scipy.sparse as sp
import torch

from . import graphs
from .csr import CsrOperator


def _read(path):
    with open(path, 'rb') as fh:
        return pickle.load(fh, encoding='latin1') if sys.version_info > (3, 0) else pickle.load(fh)


def row_normalize(m):
    """D^-1 A on a scipy matrix (propagation.py:30-37), zero rows stay zero."""
    m = sp.csr_matrix(m, dtype=np.float64)
    deg = np.asarray(m.sum(1)).reshape(-1).astype(np.float32)
    inv = np.zeros_like(deg)
    inv[deg != 0] = 1.0 / deg[deg != 0]
    return sp.diags(inv.astype(np.float64)) @ m


def assemble(x, y, tx, ty, allx, ally, graph, test_idx_reorder, dataset_name, alpha):
    test_idx_range = np.sort(test_idx_reorder)
    if dataset_name == 'citeseer':                      # isolated test nodes are missing from tx (utils.py:140-149)
        full = range(min(test_idx_reorder), max(test_idx_reorder) + 1)
        tx_ext = sp.lil_matrix((len(full), x.shape[1]))
        tx_ext[test_idx_range - min(test_idx_range), :] = tx
        tx = tx_ext
        ty_ext = np.zeros((len(full), y.shape[1]))
        ty_ext[test_idx_range - min(test_idx_range), :] = ty
        ty = ty_ext
    features = sp.vstack((allx, tx)).tolil()
    features[test_idx_reorder, :] = features[test_idx_range, :]
    labels = np.vstack((ally, ty))
    labels[test_idx_reorder, :] = labels[test_idx_range, :]
    rows, cols = [], []
    for r in graph:
        for c in graph.get(r):
            rows.append(r)
            cols.append(c)
    n = features.shape[0]
    adj = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    adj = adj + adj.T                                   # directed citations -> undirected
    adj.data[:] = 1.0                                   # adj[adj > 1] = 1
    op = graphs.zipf_smoothing_alpha(adj, alpha)        # propagation.py:91-103
    return {'adj': op, 'features': sp.csr_matrix(row_normalize(features)).astype(np.float32),
            'labels': labels.argmax(1).astype(np.int64), 'idx_train': np.arange(len(y)),
            'idx_val': np.arange(len(y), len(y) + 500), 'idx_test': test_idx_range}


def load_data(dataset_name='cora', alpha=0.5, data_dir='data', device=None):
    name = dataset_name.lower()
    parts = [_read(os.path.join(data_dir, name, 'ind.%s.%s' % (name, p))) for p in ('x', 'y', 'tx', 'ty', 'allx', 'ally', 'graph')]
    reorder = np.loadtxt(os.path.join(data_dir, name, 'ind.%s.test.index' % name), dtype=np.int64)
    d = assemble(*parts, reorder, name, alpha)
    return to_tensors(d, device)


def to_tensors(d, device=None):
    adj = CsrOperator.from_scipy(d['adj'], device)
    feats = torch.from_numpy(np.asarray(d['features'].todense(), dtype=np.float32))
    out = [feats, torch.from_numpy(d['labels']), torch.from_numpy(np.asarray(d['idx_train'], dtype=np.int64)),
           torch.from_numpy(np.asarray(d['idx_val'], dtype=np.int64)), torch.from_numpy(np.asarray(d['idx_test'], dtype=np.int64))]
    if device is not None:
        out = [t.to(device) for t in out]
    return (adj,) + tuple(out)
