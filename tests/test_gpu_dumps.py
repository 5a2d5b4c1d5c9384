"""The result files the drivers' `--dump` writes keep the reference's formats, so that
`summarize_result.py`-style tooling written against the reference reads them unchanged:
  * dynamics drivers: a torch-saved dict with the keys of heat_dynamics.py:300-311, filled as :360-368 / :434-438,
    at results/<kind>/<network>/result_<MMDD-HHMMSS>.<baseline>;
  * dgnn: a text file - line 1 str(vars(args)), line 2 the header 'Time\\tLoss\\tAccuracy\\tStep', then one
    '{:.5f}\\t{:.5f}\\t{:.5f}\\t{:.5f}' row per --iter (dgnn.py:240-244,259-261)."""
import ast
import glob
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

# heat_dynamics.py:300-311
DYNAMICS_KEYS = {'args', 'v_iter', 'abs_error', 'rel_error', 'true_y', 'predict_y', 'abs_error2', 'rel_error2', 'predict_y2',
                 'model_state_dict', 'total_time'}
# neural_dynamics.py:123-148 (measured): the NDCN state_dict
NDCN_KEYS = ['input_layer.0.weight', 'input_layer.0.bias', 'input_layer.2.weight', 'input_layer.2.bias',
             'neural_dynamic_layer.odefunc.wt.weight', 'neural_dynamic_layer.odefunc.wt.bias', 'output_layer.weight',
             'output_layer.bias']
# the flags of heat_dynamics.py:19-64 (dump_appendix is commented out there) and dgnn.py:24-70
DYNAMICS_FLAGS = {'method', 'rtol', 'atol', 'lr', 'weight_decay', 'dropout', 'hidden', 'time_tick', 'sampled_time', 'niters',
                  'test_freq', 'viz', 'gpu', 'adjoint', 'n', 'sparse', 'network', 'layout', 'seed', 'T', 'operator', 'baseline',
                  'dump'}
DGNN_FLAGS = {'no_cuda', 'fastmode', 'seed', 'epochs', 'rtol', 'atol', 'lr', 'weight_decay', 'nHiddenLayers', 'hidden',
              'dropout', 'dataset', 'model', 'iter', 'dump', 'delta', 'sms', 'normalize', 'Euler', 'T', 'time_tick',
              'no_control', 'method', 'alpha'}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


@pytest.mark.parametrize('kind,sampled', [('heat', 'equal'), ('gene', 'irregular')])
def test_dynamics_dump_has_the_reference_layout(dev, kind, sampled, tmp_path, monkeypatch, capsys):
    from ndcn_amd.drivers.dynamics import main
    monkeypatch.chdir(tmp_path)
    main(kind, ['--network', 'grid', '--sampled_time', sampled, '--baseline', 'ndcn', '--gpu', '0', '--niters', '20',
                '--test_freq', '10', '--time_tick', '20', '--method', 'euler', '--dump'])
    text = capsys.readouterr().out
    files = glob.glob(str(tmp_path / 'results' / kind / 'grid' / 'result_*.ndcn'))
    assert len(files) == 1 and re.fullmatch(r'result_\d{4}-\d{6}\.ndcn', os.path.basename(files[0]))
    assert 'Dump results as: results/%s/grid/%s' % (kind, os.path.basename(files[0])) in text
    d = torch.load(files[0], weights_only=False)
    assert set(d) == DYNAMICS_KEYS
    assert isinstance(d['args'], dict) and DYNAMICS_FLAGS <= set(d['args']) and d['args']['dump'] is True
    n_eval = 2                                                       # iterations 10 and 20 (test_freq 10)
    assert d['v_iter'] == [10, 20]
    n_nodes, n_ticks = 400, 20 if sampled == 'equal' else 24         # irregular: int(time_tick * 1.2) sampled times
    assert len(d['true_y']) == 1 and tuple(d['true_y'][0].shape) == (n_nodes, n_ticks)
    n_test = n_ticks - int(20 * 0.8) if sampled == 'equal' else None
    for key in ('abs_error', 'rel_error'):
        assert len(d[key]) == n_eval and all(isinstance(v, float) for v in d[key])
    assert len(d['predict_y']) == n_eval and all(p.shape[0] == n_nodes for p in d['predict_y'])
    if sampled == 'equal':
        assert all(tuple(p.shape) == (n_nodes, n_test) for p in d['predict_y'])
        assert d['abs_error2'] == [] and d['rel_error2'] == [] and d['predict_y2'] == []      # no extrapolation split
    else:
        assert len(d['abs_error2']) == len(d['rel_error2']) == len(d['predict_y2']) == n_eval
    assert len(d['model_state_dict']) == n_eval and list(d['model_state_dict'][0]) == NDCN_KEYS
    assert isinstance(d['total_time'], float) and d['total_time'] > 0
    # a consumer can rebuild the model from a dumped state_dict
    from ndcn_amd.neural_dynamics import NDCN
    m = NDCN(1, d['args']['hidden'], None, 1, method='euler')
    m.load_state_dict(d['model_state_dict'][-1])


def test_dgnn_dump_is_the_reference_tsv(dev, tmp_path, monkeypatch):
    import scipy.sparse as sp
    from ndcn_amd import CsrOperator
    from ndcn_amd.drivers import dgnn
    monkeypatch.chdir(tmp_path)
    dd, g = load_golden('dataset_cora'), load_golden('operators_cora')
    n = int(g['n'])
    adj = CsrOperator.from_arrays(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n), dev)
    feats = sp.csr_matrix((dd['feat_data'], dd['feat_indices'].astype(np.int64), dd['feat_indptr']), shape=tuple(dd['feat_shape']))
    data = (adj, torch.from_numpy(feats.toarray()).to(dev), torch.from_numpy(dd['labels'].astype(np.int64)).to(dev),
            torch.from_numpy(dd['idx_train'].astype(np.int64)).to(dev), torch.from_numpy(dd['idx_val'].astype(np.int64)).to(dev),
            torch.from_numpy(dd['idx_test'].astype(np.int64)).to(dev))
    accs = dgnn.main(['--dataset', 'cora', '--model', 'differential_gcn', '--iter', '2', '--dropout', '0', '--hidden', '32',
                      '--T', '1.2', '--time_tick', '4', '--epochs', '5', '--no_control', '--alpha', '0', '--seed', '0', '--dump'],
                     data=data, quiet=True)
    files = glob.glob(str(tmp_path / 'results' / 'results_*.txt'))
    assert len(files) == 1 and ':' not in os.path.basename(files[0])          # dgnn.py:241: colons of the timestamp replaced
    lines = open(files[0]).read().splitlines()
    args = ast.literal_eval(lines[0])                                          # vars(args).__str__()
    assert isinstance(args, dict) and DGNN_FLAGS <= set(args) and args['dump'] is True and args['iter'] == 2
    assert lines[1] == 'Time\tLoss\tAccuracy\tStep'
    rows = lines[2:]
    assert len(rows) == 2
    for row, acc in zip(rows, accs):
        cells = row.split('\t')
        assert len(cells) == 4 and all(re.fullmatch(r'-?\d+\.\d{5}', c) for c in cells)
        assert abs(float(cells[2]) - float(acc)) < 1e-5 and float(cells[3]) == 0.0   # dgnn.py:258: time_step = 0
