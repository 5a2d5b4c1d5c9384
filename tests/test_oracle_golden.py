"""Pins the CPU oracle (oracle/ndcn_oracle.py) to the fixtures captured from the reference
(tools/gen_golden.py).  CPU-only."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, load_golden
from oracle import ndcn_oracle as orc

torch.set_num_threads(1)
TOL = 1e-6     # restatement checked against the reference to <= 1e-6


def T(a):
    return torch.from_numpy(np.asarray(a))


def op_from(d, layout='coo', prefix=''):
    shape = d['shape'] if 'shape' in d else (int(d['n']),) * 2
    args = (d[prefix + 'indptr'], d[prefix + 'indices'], d[prefix + 'data'], shape)
    return orc.coo_from_csr(*args) if layout == 'coo' else orc.dense_from_csr(*args)


def names(pattern):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern)))


@pytest.mark.parametrize('name', names('rhs_*.npz'))
def test_rhs(name):
    d = load_golden(name)
    layout = 'coo' if name.endswith('_coo') else 'dense'
    out = orc.odefunc_rhs(op_from(d, layout), T(d['x']), T(d['W']), T(d['b']),
                          no_graph='no_graph' in name, no_control='no_control' in name)
    assert np.abs(out.numpy() - d['out']).max() <= TOL


@pytest.mark.parametrize('name', names('fixed_*.npz'))
def test_fixed_grid(name):
    d = load_golden(name)
    method = name.split('_')[1]
    f = orc.OracleODEFunc(op_from(d), T(d['W']), T(d['b']))
    y = orc.odeint(f, T(d['x0']), T(d['t']), method=method)
    assert y.shape == d['traj'].shape
    assert np.array_equal(y[0].numpy(), d['x0'])          # out[0] is y0 itself (solvers.py:86)
    assert np.abs(y.numpy() - d['traj']).max() <= TOL
    per = {'euler': 1, 'midpoint': 2, 'rk4': 4}[method]
    assert f.nfe == per * (len(d['t']) - 1)


@pytest.mark.parametrize('name', names('dopri5_*.npz'))
def test_dopri5(name):
    d = load_golden(name)
    f = orc.OracleODEFunc(op_from(d), T(d['W']), T(d['b']), no_control='no_control' in name)
    log = []
    opts = {k[4:]: float(v) for k, v in d.items() if k.startswith('opt_')} or None
    y = orc.odeint(f, T(d['x0']), T(d['t']), rtol=float(d['rtol']), atol=float(d['atol']), method='dopri5', step_log=log,
                   options=opts)
    ref = d['steplog']
    assert f.nfe == int(d['nfe']) == 2 + 6 * len(ref)
    log = np.array(log)
    assert log.shape == ref.shape
    assert np.array_equal(log[:, 2], ref[:, 2])            # identical accept / reject sequence
    assert np.allclose(log[:, [0, 1, 4]], ref[:, [0, 1, 4]], rtol=1e-6, atol=0)
    assert np.allclose(log[:, 3], ref[:, 3], rtol=1e-4, atol=1e-12)
    scale = max(1.0, np.abs(d['traj']).max())
    assert np.abs(y.numpy() - d['traj']).max() <= TOL * scale


@pytest.mark.parametrize('name', names('adams_*.npz'))
def test_adams(name):
    """The oracle's variable-coefficient Adams restatement against the reference's own run (adams.py:62-170): the
    trajectory bit for bit, and every attempted step's (t_n, next_t, order, accepted, following next_t)."""
    d = load_golden(name)
    f = orc.OracleODEFunc(op_from(d), T(d['W']), T(d['b']), no_control=bool(d['no_control']))
    log = []
    opts = {k[4:]: (int(v) if k == 'opt_max_order' else float(v)) for k, v in d.items() if k.startswith('opt_')} or None
    y = orc.odeint(f, T(d['x0']), T(d['t']), rtol=float(d['rtol']), atol=float(d['atol']), method='adams', step_log=log,
                   options=opts)
    ref, log = d['steplog'], np.array(log)
    assert f.nfe == int(d['nfe']) and log.shape[0] == ref.shape[0]
    assert np.array_equal(log[:, [0, 1, 2, 3, 5]], ref)
    assert np.array_equal(y.numpy(), d['traj'])
    assert len(set(ref[:, 2])) >= 2                        # the order really varies


@pytest.mark.parametrize('name', names('ndcn_*.npz'))
def test_ndcn_end_to_end(name):
    d = load_golden(name)
    variant, method = name[len('ndcn_'):].rsplit('_', 1)
    sd = {k[4:].replace('__', '.'): T(v) for k, v in d.items() if k.startswith('sd__')}
    out = orc.ndcn_forward(sd, op_from(d, 'dense'), T(d['t']), T(d['x0']), method,
                           no_embed=variant == 'no_embed', no_graph=variant == 'no_graph',
                           no_control=variant == 'no_control')
    scale = max(1.0, np.abs(d['out']).max())
    assert np.abs(out.numpy() - d['out']).max() <= TOL * scale


@pytest.mark.parametrize('name', names('truth_*.npz'))
def test_truth_dynamics(name):
    d = load_golden(name)
    n = int(d['n'])
    if name.endswith('coo'):
        mk = lambda p: orc.coo_from_csr(d[p + 'indptr'], d[p + 'indices'], d[p + 'data'], (n, n))
    else:
        mk = lambda p: orc.dense_from_csr(d[p + 'indptr'], d[p + 'indices'], d[p + 'data'], (n, n))
    A, L = mk('A_'), mk('L_')
    if 'heat' in name:
        f = lambda t, x: orc.heat_rhs(L, x)
    elif 'gene' in name:
        f = lambda t, x: orc.gene_rhs(A, x)
    else:
        f = lambda t, x: orc.mutual_rhs(A, x)
    y = orc.odeint(f, T(d['x0']), T(d['t']), method='dopri5')      # odeint defaults rtol 1e-7 / atol 1e-9
    assert np.abs(y.numpy() - d['traj']).max() <= 2e-5 * max(1.0, np.abs(d['traj']).max())


def test_mutual_edgewise_equals_dense_branch():
    d = load_golden('truth_mutual_coo')
    n = int(d['n'])
    A = orc.coo_from_csr(d['A_indptr'], d['A_indices'], d['A_data'], (n, n))
    x = T(d['traj'][3])
    dense = orc.mutual_rhs(A, x).numpy()
    edge = orc.mutual_rhs_edgewise(d['A_indptr'], d['A_indices'], d['A_data'], x.numpy())
    assert np.abs(dense - edge).max() <= 1e-4 * np.abs(dense).max()


def test_heat_closed_form_known_answer():
    # K1: dopri5 truth vs V exp(-Lambda t) V^T x0 (8.4e-7 measured on the reference)
    d = load_golden('truth_heat_dense')
    n = int(d['n'])
    L = orc.dense_from_csr(d['L_indptr'], d['L_indices'], d['L_data'], (n, n)).numpy()
    exact = orc.heat_closed_form(L, d['x0'], d['t'])
    assert np.abs(exact - d['traj']).mean() < 5e-6
    # K2: total heat is conserved by -L
    assert abs(d['traj'][-1].sum() - d['x0'].sum()) < 1e-2


def test_operator_builders():
    d = load_golden('operators_grid400')
    A = orc.grid_8_neighbor_dense(20)
    ref_A = sp.csr_matrix((d['A_data'], d['A_indices'], d['A_indptr']), shape=(400, 400)).toarray()
    assert np.array_equal(A, ref_A)
    for kind, fn in (('norm_lap', orc.normalized_laplacian_dense), ('norm_adj', orc.normalized_adj_dense),
                     ('kipf', orc.zipf_smoothing_dense), ('lap', orc.laplacian_dense)):
        ref = sp.csr_matrix((d[kind + '_data'], d[kind + '_indices'], d[kind + '_indptr']), shape=(400, 400)).toarray()
        assert np.abs(fn(A).astype(np.float32) - ref).max() <= 1e-7, kind
    assert np.array_equal(orc.x0_blocks(20), d['x0'])
    d5 = load_golden('operators_grid25')
    assert np.array_equal(orc.grid_8_neighbor_dense(5),
                          sp.csr_matrix((d5['A_data'], d5['A_indices'], d5['A_indptr']), shape=(25, 25)).toarray())


@pytest.mark.parametrize('name', ['cora', 'pubmed'])
def test_zipf_alpha_operator(name):
    d = load_golden('operators_' + name)
    n = int(d['n'])
    adj = sp.csr_matrix((np.ones(len(d['adj_indices'])), d['adj_indices'], d['adj_indptr']), shape=(n, n))
    for alpha, tag in ((0.0, 'alpha00'), (0.5, 'alpha05')):
        op = orc.zipf_smoothing_alpha(adj, alpha).astype(np.float32)
        ref = sp.csr_matrix((d[tag + '_data'], d[tag + '_indices'], d[tag + '_indptr']), shape=(n, n))
        assert abs(op - ref).max() <= 1e-6


@pytest.mark.parametrize('name,H', [('cora', 64), ('pubmed', 16)])
def test_dgnn_block(name, H):
    d = load_golden('dgnn_%s_H%d' % (name, H))
    g = load_golden('operators_' + name)
    n = int(g['n'])
    A = orc.coo_from_csr(g['alpha00_indptr'], g['alpha00_indices'], g['alpha00_data'], (n, n))
    f = orc.OracleODEFunc(A, None, None, no_control=True)
    log = []
    y = orc.odeint(f, T(d['x']), T(d['t']), rtol=.1, atol=.1, method='dopri5', step_log=log)[-1]
    assert f.nfe == int(d['nfe'])
    assert np.array_equal(np.array(log)[:, 2], d['steplog'][:, 2])
    assert np.abs(y.numpy() - d['out']).max() <= 1e-5


def test_rownorm_resblock_gcn_resgcn():
    """G10: RowNorm, ResBlock in its four configurations, models.GCN and dgnn's resGCN Sequential (eval mode)."""
    x = T(load_golden('resgcn_rownorm')['x'])
    assert torch.equal(orc.row_normalization(x.clone()), T(load_golden('resgcn_rownorm')['out']))
    for tag in ('plain', 'norm', 'tv', 'euler', 'norm_tv'):
        d = load_golden('resgcn_block_' + tag)
        out = orc.resblock(op_from(d), x.clone(), T(d['W']) if 'W' in d else None, T(d['b']) if 'b' in d else None,
                           time_step=T(d['time_step']) if 'time_step' in d else 1.0, normalize='norm' in tag)
        assert (out - T(d['out'])).abs().max() <= TOL, tag
    d = load_golden('resgcn_gcn')
    sd = {k[3:]: T(v) for k, v in d.items() if k.startswith('sd_')}
    assert (orc.gcn_forward(sd, op_from(d), T(d['x']), n_middle=1) - T(d['out'])).abs().max() <= TOL
    for tag, norm in (('model', False), ('model_norm_euler', True)):
        d = load_golden('resgcn_' + tag)
        sd = {k[3:]: T(v) for k, v in d.items() if k.startswith('sd_')}
        assert (orc.resgcn_forward(sd, op_from(d), T(d['x']), 2, normalize=norm) - T(d['out'])).abs().max() <= TOL, tag


def test_input_checks():
    f = lambda t, x: -x
    y0 = torch.ones(3)
    with pytest.raises(AssertionError):
        orc.odeint(f, y0, torch.tensor([0., 1., 0.5]), method='euler')
    with pytest.raises(TypeError):
        orc.odeint(f, torch.ones(3, dtype=torch.int64), torch.tensor([0., 1.]), method='euler')
    with pytest.raises(TypeError):
        orc.odeint(f, y0, torch.tensor([0, 1]), method='euler')
    with pytest.raises(ValueError):
        orc.odeint(f, y0, torch.tensor([0., 1.]), options={'step_size': 0.1})
    with pytest.raises(KeyError):
        orc.odeint(f, y0, torch.tensor([0., 1.]), method='nope')
    with pytest.raises(AssertionError):
        orc.odeint(f, [y0], torch.tensor([0., 1.]), method='euler')
