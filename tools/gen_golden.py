#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ by RUNNING the reference.

Dev-only tool.  It contains no reference code: it puts /root/reference on sys.path,
imports the reference's own modules (neural_dynamics, torchdiffeq, utils_in_learn_dynamics,
propagation) and the module-level set-up of its three dynamics scripts, feeds them seeded
inputs, and stores (inputs, expected outputs) as small .npz files.  Only those data files
travel to the GPU box; the reference's Python never does.  When /root/reference is absent
(GPU box) it exits cleanly without touching anything.

Fixture families:
  G1 rhs_*.npz        ODEFunc.forward                       neural_dynamics.py:20-39
  G2 fixed_*.npz      odeint euler / midpoint / rk4         torchdiffeq/_impl/{solvers,fixed_grid,rk_common}.py
  G3 dopri5_*.npz     odeint dopri5 + per-attempt step log  torchdiffeq/_impl/{dopri5,interp,misc}.py
  G4 ndcn_*.npz       NDCN end to end (state_dict + output) neural_dynamics.py:122-160
  G5 truth_*.npz      heat / gene / mutualistic truth       {heat,gene,mutualistic}_dynamics.py:186-232
  G6 operators_*.npz  dense operator builders, zipf alpha   utils_in_learn_dynamics.py:80-157, propagation.py:91-103
  G7 dgnn_*.npz       ODEBlock2(no_control, terminal)       dgnn.py:173-182 on the Planetoid topologies
  G10 resgcn_*.npz    RowNorm / ResBlock / GCN / resGCN     ode_gcn.py:9-60, models.py:8-47, dgnn.py:129-140
  G11 layout_*.npz    generate_node_mapping degree/community utils_in_learn_dynamics.py:212-230 (+ the P A P^T of :233-247)
  G12 gconv_dense.npz GraphConvolution (dense A, flattened)  neural_dynamics.py:163-176
  G13 adams_*.npz     odeint adams + per-attempt step log    torchdiffeq/_impl/adams.py:62-170
"""
import os
import sys
import io
import pickle
import runpy
import contextlib

REF = '/root/reference'
if not os.path.isdir(REF):
    print('gen_golden: %s not present - nothing to do' % REF)
    sys.exit(0)

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, REF)
os.environ.setdefault('MPLBACKEND', 'Agg')
import neural_dynamics as ref_nd            # noqa: E402  (the reference)
import torchdiffeq as ref_ode               # noqa: E402  (the reference's vendored copy)
import torchdiffeq._impl.dopri5 as ref_dopri5   # noqa: E402
import utils_in_learn_dynamics as ref_u     # noqa: E402
import propagation as ref_prop              # noqa: E402

assert ref_ode.__file__.startswith(REF), ref_ode.__file__

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)      # deterministic summation order inside ATen


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **conv)
    print('wrote %-34s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def csr_of(dense):
    m = sp.csr_matrix(np.asarray(dense, dtype=np.float32))
    m.sort_indices()
    return dict(indptr=m.indptr.astype(np.int32), indices=m.indices.astype(np.int32),
                data=m.data.astype(np.float32), shape=np.array(m.shape, dtype=np.int64))


def grid_operator(S, kind='norm_lap'):
    A = ref_u.grid_8_neighbor_graph(S)
    if kind == 'norm_lap':
        OM = torch.FloatTensor(ref_u.normalized_laplacian(A.numpy()))
    elif kind == 'norm_adj':
        OM = torch.FloatTensor(ref_u.normalized_adj(A.numpy()))
    elif kind == 'kipf':
        OM = torch.FloatTensor(ref_u.zipf_smoothing(A.numpy()))
    elif kind == 'lap':
        OM = torch.diag(A.sum(1)) - A
    return A, OM


def x0_blocks(S):
    # the reference's initial image, heat_dynamics.py:178-182
    x0 = torch.zeros(S, S)
    x0[int(0.05 * S):int(0.25 * S), int(0.05 * S):int(0.25 * S)] = 25
    x0[int(0.45 * S):int(0.75 * S), int(0.45 * S):int(0.75 * S)] = 20
    x0[int(0.05 * S):int(0.25 * S), int(0.35 * S):int(0.65 * S)] = 17
    return x0.view(-1, 1).float()


# ----------------------------------------------------------------------------- G1
def gen_rhs():
    _, OM = grid_operator(20)
    OMs = ref_u.torch_sensor_to_torch_sparse_tensor(OM)
    for H in (1, 20, 256):
        for flag in ('default', 'no_control', 'no_graph'):
            for layout in ('dense', 'coo'):
                if H == 256 and layout == 'dense':
                    continue            # keep the fixture set small
                torch.manual_seed(100 + H)
                f = ref_nd.ODEFunc(H, OM if layout == 'dense' else OMs, dropout=0.0,
                                   no_graph=(flag == 'no_graph'), no_control=(flag == 'no_control'))
                x = torch.randn(400, H)
                with torch.no_grad():
                    out = f(torch.tensor(0.0), x)
                save('rhs_grid400_H%d_%s_%s' % (H, flag, layout), x=x, W=f.wt.weight, b=f.wt.bias,
                     out=out, **csr_of(OM))


# ----------------------------------------------------------------------------- G2
def make_func(H, OM, seed, **kw):
    torch.manual_seed(seed)
    f = ref_nd.ODEFunc(H, OM, dropout=0.0, **kw)
    x = torch.rand(OM.shape[0], H)
    return f, x


def gen_fixed():
    _, OM = grid_operator(20)
    OMs = ref_u.torch_sensor_to_torch_sparse_tensor(OM)
    t_eq = torch.linspace(0., 5., 21)
    rng = np.random.RandomState(7)
    t_ir = torch.linspace(0., 5., 200)[np.sort(rng.permutation(200)[:14])].clone()
    t_ir[0] = 0
    for method in ('euler', 'midpoint', 'rk4'):
        for tname, t in (('equal', t_eq), ('irregular', t_ir)):
            f, x = make_func(20, OMs, 11)
            with torch.no_grad():
                y = ref_ode.odeint(f, x, t, method=method)
            save('fixed_%s_%s' % (method, tname), x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, **csr_of(OM))
    # decreasing time vector (misc.py:184-187) on rk4
    f, x = make_func(20, OMs, 11)
    t = torch.linspace(1., 0., 6)
    with torch.no_grad():
        y = ref_ode.odeint(f, x, t, method='rk4')
    save('fixed_rk4_decreasing', x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, **csr_of(OM))


# ----------------------------------------------------------------------------- G3
class StepLog:
    """Records every attempted dopri5 step by wrapping (not editing) the reference's methods."""

    def __init__(self):
        self.rows = []
        self.nfe = 0
        self._ratio = None

    @contextlib.contextmanager
    def attach(self):
        orig_step = ref_dopri5.Dopri5Solver._adaptive_dopri5_step
        orig_ratio = ref_dopri5._compute_error_ratio
        log = self

        def ratio_wrap(*a, **k):
            r = orig_ratio(*a, **k)
            log._ratio = float(r[0])
            return r

        def step_wrap(solver, rk_state):
            t0, dt = float(rk_state.t1), float(rk_state.dt)
            new = orig_step(solver, rk_state)
            accepted = float(new.t1) != t0
            log.rows.append((t0, dt, 1.0 if accepted else 0.0, log._ratio, float(new.dt)))
            return new

        ref_dopri5.Dopri5Solver._adaptive_dopri5_step = step_wrap
        ref_dopri5._compute_error_ratio = ratio_wrap
        try:
            yield self
        finally:
            ref_dopri5.Dopri5Solver._adaptive_dopri5_step = orig_step
            ref_dopri5._compute_error_ratio = orig_ratio


class CountingFunc(torch.nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f
        self.nfe = 0

    def forward(self, t, x):
        self.nfe += 1
        return self.f(t, x)


def gen_dopri5():
    _, OM = grid_operator(20)
    OMs = ref_u.torch_sensor_to_torch_sparse_tensor(OM)
    cases = [('loose', .01, .001, torch.linspace(0., 5., 21)),
             ('dgnn', .1, .1, torch.linspace(0., 1.2, 16)),
             ('tight', 1e-7, 1e-9, torch.linspace(0., 2., 6)),
             ('twotick', .01, .001, torch.tensor([0., 5.]))]
    for name, rtol, atol, t in cases:
        f, x = make_func(20, OMs, 23)
        cf = CountingFunc(f)
        log = StepLog()
        with torch.no_grad(), log.attach():
            y = ref_ode.odeint(cf, x, t, rtol=rtol, atol=atol, method='dopri5')
        save('dopri5_%s' % name, x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, rtol=rtol, atol=atol,
             steplog=np.array(log.rows, dtype=np.float64), nfe=cf.nfe, **csr_of(OM))
    # non-default controller options (dopri5.py:60,72-74): small ifactor / large dfactor / tighter safety, tolerances
    # that produce rejected steps
    f, x = make_func(20, OMs, 37)
    cf = CountingFunc(f)
    log = StepLog()
    t = torch.linspace(0., 4., 9)
    opts = dict(safety=0.99, ifactor=20.0, dfactor=0.5)
    with torch.no_grad(), log.attach():
        y = ref_ode.odeint(cf, x, t, rtol=1e-3, atol=1e-4, method='dopri5', options=dict(opts))
    # the controller itself on (dt, ratio) pairs that hit every branch (misc.py:160-170), with the solver's own
    # conversion of the option values (dopri5.py:72-74)
    import torchdiffeq._impl.misc as ref_misc
    cv = lambda v: ref_misc._convert_to_tensor(v, dtype=torch.float64)
    ctl_in = [(0.25, 0.0), (0.25, 1e-9), (0.25, 0.3), (0.25, 0.999), (0.25, 1.0), (0.25, 1.7), (0.25, 40.0), (0.25, 1e9)]
    ctl_out = [float(ref_misc._optimal_step_size(torch.tensor(dt, dtype=torch.float64), [torch.tensor(r, dtype=torch.float32)],
                                                 safety=cv(opts['safety']), ifactor=cv(opts['ifactor']), dfactor=cv(opts['dfactor'])))
               for dt, r in ctl_in]
    save('dopri5_options', x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, rtol=1e-3, atol=1e-4,
         steplog=np.array(log.rows, dtype=np.float64), nfe=cf.nfe, ctl_in=np.array(ctl_in), ctl_out=np.array(ctl_out),
         **{'opt_' + k: v for k, v in opts.items()}, **csr_of(OM))
    # no_control (pure SpMM+ReLU RHS) and a decreasing time vector
    f, x = make_func(20, OMs, 29, no_control=True)
    cf = CountingFunc(f)
    log = StepLog()
    t = torch.linspace(0., 3., 7)
    with torch.no_grad(), log.attach():
        y = ref_ode.odeint(cf, x, t, rtol=.01, atol=.001, method='dopri5')
    save('dopri5_no_control', x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, rtol=.01, atol=.001,
         steplog=np.array(log.rows, dtype=np.float64), nfe=cf.nfe, **csr_of(OM))
    f, x = make_func(20, OMs, 31)
    cf = CountingFunc(f)
    log = StepLog()
    t = torch.linspace(2., 0., 5)
    with torch.no_grad(), log.attach():
        y = ref_ode.odeint(cf, x, t, rtol=.01, atol=.001, method='dopri5')
    save('dopri5_decreasing', x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, rtol=.01, atol=.001,
         steplog=np.array(log.rows, dtype=np.float64), nfe=cf.nfe, **csr_of(OM))


# ----------------------------------------------------------------------------- G4
def gen_ndcn():
    _, OM = grid_operator(20)
    x0 = x0_blocks(20)
    t = torch.linspace(0., 5., 21)
    variants = {
        'ndcn': dict(hidden_size=20, no_embed=False, no_graph=False, no_control=False),
        'no_embed': dict(hidden_size=1, no_embed=True, no_graph=False, no_control=False),
        'no_control': dict(hidden_size=20, no_embed=False, no_graph=False, no_control=True),
        'no_graph': dict(hidden_size=20, no_embed=False, no_graph=True, no_control=False),
    }
    for name, kw in variants.items():
        for method in ('euler', 'dopri5'):
            torch.manual_seed(0)
            m = ref_nd.NDCN(input_size=1, A=OM, num_classes=1, dropout=0.0, rtol=.01, atol=.001, method=method, **kw)
            with torch.no_grad():
                y = m(t, x0)
            sd = {'sd__' + k.replace('.', '__'): v for k, v in m.state_dict().items()}
            save('ndcn_%s_%s' % (name, method), x0=x0, t=t, out=y, **sd, **csr_of(OM))


# ----------------------------------------------------------------------------- G5
def run_script_setup(script, argv):
    """Execute a reference driver's module-level set-up (graph, operator, truth solve, model build).
    Its training loop sits under `if __name__ == '__main__'` and is not entered."""
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [script] + argv
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            g = runpy.run_path(os.path.join(REF, script), run_name='ref_setup')
    finally:
        sys.argv, _ = old_argv, os.chdir(old_cwd)
    return g


def gen_truth():
    for script, tag in (('heat_dynamics.py', 'heat'), ('gene_dynamics.py', 'gene'), ('mutualistic_dynamics.py', 'mutual')):
        for sparse in (False, True):
            argv = ['--network', 'grid', '--sampled_time', 'equal', '--baseline', 'ndcn', '--gpu', '-1',
                    '--time_tick', '21'] + (['--sparse'] if sparse else [])
            g = run_script_setup(script, argv)
            A = g['A'].to_dense() if g['A'].is_sparse else g['A']
            L = g['L'].to_dense() if g['L'].is_sparse else g['L']
            OM = g['OM'].to_dense() if g['OM'].is_sparse else g['OM']
            a, l, om = csr_of(A), csr_of(L), csr_of(OM)
            save('truth_%s_%s' % (tag, 'coo' if sparse else 'dense'), x0=g['x0'], t=g['t'], traj=g['solution_numerical'],
                 A_indptr=a['indptr'], A_indices=a['indices'], A_data=a['data'],
                 L_indptr=l['indptr'], L_indices=l['indices'], L_data=l['data'],
                 OM_indptr=om['indptr'], OM_indices=om['indices'], OM_data=om['data'], n=np.int64(A.shape[0]))


# ----------------------------------------------------------------------------- G6
def load_planetoid_adj(name):
    """Adjacency of a Planetoid pickle, symmetrised as utils.py:183-196 does (utils.load_data itself
    no longer runs under scipy 1.15)."""
    with open(os.path.join(REF, 'data', name, 'ind.%s.graph' % name), 'rb') as fh:
        graph = pickle.load(fh, encoding='latin1')
    rows, cols = [], []
    for r in graph:
        for c in graph.get(r):
            rows.append(r)
            cols.append(c)
    n = max(max(rows), max(cols)) + 1
    adj = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    adj = adj + adj.T
    adj[adj > 1] = 1
    return adj.tocsr()


def gen_operators():
    A, _ = grid_operator(20)
    a = csr_of(A)
    arrs = dict(A_indptr=a['indptr'], A_indices=a['indices'], A_data=a['data'], n=np.int64(400))
    for kind in ('norm_lap', 'norm_adj', 'kipf', 'lap'):
        _, OM = grid_operator(20, kind)
        c = csr_of(OM)
        arrs.update({kind + '_indptr': c['indptr'], kind + '_indices': c['indices'], kind + '_data': c['data']})
    arrs['x0'] = x0_blocks(20)
    save('operators_grid400', **arrs)
    # non-square canvas: n=10 nodes asked, ceil(sqrt) = 4 (heat_dynamics.py:81)
    A5, _ = grid_operator(5)
    c = csr_of(A5)
    save('operators_grid25', A_indptr=c['indptr'], A_indices=c['indices'], A_data=c['data'], n=np.int64(25))
    for name in ('cora', 'pubmed'):
        adj = load_planetoid_adj(name)
        adj.sort_indices()
        arrs = dict(adj_indptr=adj.indptr.astype(np.int32), adj_indices=adj.indices.astype(np.int32),
                    n=np.int64(adj.shape[0]))
        for alpha in (0.0, 0.5):
            with contextlib.redirect_stdout(io.StringIO()):
                op = ref_prop.Propagation(adj).zipf_smoothing_alpha(alpha)
            op = sp.csr_matrix(op).astype(np.float32)
            op.sort_indices()
            tag = 'alpha%02d' % int(alpha * 10)
            arrs.update({tag + '_indptr': op.indptr.astype(np.int32), tag + '_indices': op.indices.astype(np.int32),
                         tag + '_data': op.data.astype(np.float32)})
        save('operators_%s' % name, **arrs)


# ----------------------------------------------------------------------------- G7
def gen_dgnn():
    sys.path.insert(0, REF)
    import utils as ref_utils
    for name, H in (('cora', 64), ('pubmed', 16)):
        adj = load_planetoid_adj(name)
        with contextlib.redirect_stdout(io.StringIO()):
            op = ref_prop.Propagation(adj).zipf_smoothing_alpha(0.0)
        A = ref_utils.sparse_csr_matrix_to_torch_sparse_tensor(sp.csr_matrix(op))
        torch.manual_seed(5)
        x = torch.tanh(torch.randn(adj.shape[0], H))          # what Linear+Tanh would hand the block
        t = torch.linspace(0, 1.2, 16)
        f = ref_nd.ODEFunc(H, A, dropout=0.0, no_control=True)
        cf = CountingFunc(f)
        blk = ref_nd.ODEBlock2(cf, t, rtol=.1, atol=.1, method='dopri5', terminal=True)
        log = StepLog()
        with torch.no_grad(), log.attach():
            y = blk(x)
        save('dgnn_%s_H%d' % (name, H), x=x, t=t, out=y, steplog=np.array(log.rows, dtype=np.float64), nfe=cf.nfe)


# ----------------------------------------------------------------------------- G8
def gen_dataset():
    """Cora as utils.load_data (utils.py:91-230) assembles it - that function itself no longer runs under scipy 1.15
    (SURVEY 8c), so its steps are replayed here on the reference's own data files: features = vstack(allx, tx) with
    the test rows re-ordered, row-normalised (propagation.py:30-37); labels likewise; the reference's index split."""
    name = 'cora'
    objs = []
    for part in ('x', 'y', 'tx', 'ty', 'allx', 'ally'):
        with open(os.path.join(REF, 'data', name, 'ind.%s.%s' % (name, part)), 'rb') as fh:
            objs.append(pickle.load(fh, encoding='latin1'))
    x, y, tx, ty, allx, ally = objs
    test_idx_reorder = np.loadtxt(os.path.join(REF, 'data', name, 'ind.%s.test.index' % name), dtype=np.int64)
    test_idx_range = np.sort(test_idx_reorder)
    features = sp.vstack((allx, tx)).tolil()
    features[test_idx_reorder, :] = features[test_idx_range, :]
    labels = np.vstack((ally, ty))
    labels[test_idx_reorder, :] = labels[test_idx_range, :]
    with contextlib.redirect_stdout(io.StringIO()):
        fn = sp.csr_matrix(ref_prop.Propagation(sp.csr_matrix(features)).row_normalization()).astype(np.float32)
    fn.sort_indices()
    save('dataset_cora', feat_indptr=fn.indptr.astype(np.int32), feat_indices=fn.indices.astype(np.int16),
         feat_data=fn.data.astype(np.float32), feat_shape=np.array(fn.shape, dtype=np.int64),
         labels=labels.argmax(1).astype(np.int16), idx_train=np.arange(len(y), dtype=np.int32),
         idx_val=np.arange(len(y), len(y) + 500, dtype=np.int32), idx_test=test_idx_range.astype(np.int32))


# ----------------------------------------------------------------------------- G9
def gen_adjoint():
    """Gradients of the reference's odeint_adjoint (torchdiffeq/_impl/adjoint.py:105-133) on a small case."""
    _, OM = grid_operator(12)
    OMs = ref_u.torch_sensor_to_torch_sparse_tensor(OM)
    for method, rtol, atol in (('dopri5', 1e-5, 1e-7), ('rk4', 0.0, 0.0)):
        torch.manual_seed(41)
        f = ref_nd.ODEFunc(8, OMs, dropout=0.0)
        x0 = torch.rand(144, 8, requires_grad=True)
        t = torch.linspace(0., 1., 5 if method == 'dopri5' else 21)
        target = torch.rand(len(t), 144, 8)
        kw = dict(method=method) if method == 'rk4' else dict(method=method, rtol=rtol, atol=atol)
        y = ref_ode.odeint_adjoint(f, x0, t, **kw)
        loss = torch.nn.functional.l1_loss(y, target)
        loss.backward()
        save('adjoint_%s' % method, x0=x0.detach(), t=t, target=target, W=f.wt.weight.detach(), b=f.wt.bias.detach(),
             traj=y.detach(), loss=loss.detach(), g_x0=x0.grad, g_W=f.wt.weight.grad, g_b=f.wt.bias.grad,
             rtol=rtol, atol=atol, **csr_of(OM))


# ----------------------------------------------------------------------------- G10
def gen_resgcn():
    import utils as ref_utils
    import ode_gcn as ref_og
    import models as ref_models
    adj = load_planetoid_adj('cora')
    with contextlib.redirect_stdout(io.StringIO()):
        op = sp.csr_matrix(ref_prop.Propagation(adj).zipf_smoothing_alpha(0.0))
    A = ref_utils.sparse_csr_matrix_to_torch_sparse_tensor(op)
    op.sort_indices()
    csr = dict(indptr=op.indptr.astype(np.int32), indices=op.indices.astype(np.int32), data=op.data.astype(np.float32),
               shape=np.array(op.shape, dtype=np.int64))
    n, H = adj.shape[0], 32
    torch.manual_seed(11)
    x = torch.randn(n, H)
    x[17] = 0                                                  # an all-zero row: 0 / max(0, eps)
    with torch.no_grad():
        save('resgcn_rownorm', x=x, out=ref_og.RowNorm()(x.clone()))
        for tag, kw in (('plain', {}), ('norm', dict(normalize=True)), ('tv', dict(time_varying=True)),
                        ('euler', dict(Euler=True)), ('norm_tv', dict(normalize=True, time_varying=True))):
            torch.manual_seed(3)
            blk = ref_og.ResBlock(H, A, **kw).eval()
            extra = {}
            if kw.get('time_varying'):
                extra.update(W=blk.linear.weight, b=blk.linear.bias)
            if kw.get('Euler'):
                extra.update(time_step=blk.time_step)
            save('resgcn_block_%s' % tag, out=blk(x.clone()), **extra, **csr)     # input: x of resgcn_rownorm.npz
        feat = torch.rand(n, 32)
        torch.manual_seed(4)
        gcn = ref_models.GCN(32, 16, 7, dropout=0.5, num_middle_layers=1).eval()
        save('resgcn_gcn', x=feat, out=gcn(feat, A), **{'sd_' + k: v for k, v in gcn.state_dict().items()}, **csr)
        for tag, norm, euler in (('model', False, False), ('model_norm_euler', True, True)):
            torch.manual_seed(5)
            layers = [torch.nn.Linear(32, H), torch.nn.ReLU(inplace=True)]
            layers += [ref_og.ResBlock(H, A, dropout=0.5, normalize=norm, Euler=euler) for _ in range(2)]
            layers += [torch.nn.Linear(H, 7)]
            model = torch.nn.Sequential(*layers).eval()
            save('resgcn_%s' % tag, x=feat, out=model(feat), **{'sd_' + k: v for k, v in model.state_dict().items()}, **csr)


def gen_layout():
    """G11: the reference's node mapping on its own graph generators (heat_dynamics.py:87-109 at n = 400).
    networkx_reorder_nodes itself calls nx.to_scipy_sparse_matrix, removed in networkx 3: the mapping
    function is what still runs; the re-labelled adjacency new_A[map[i], map[j]] = A[i, j] is formed here with the
    two calls' modern equivalents."""
    import networkx as nx
    n, seed = 400, 0
    n1, n2, n3 = int(n / 3), int(n / 3), int(n / 4)
    graphs = {
        'random': nx.erdos_renyi_graph(n, 0.1, seed=seed),
        'power_law': nx.barabasi_albert_graph(n, 5, seed=seed),
        'small_world': nx.newman_watts_strogatz_graph(400, 5, 0.5, seed=seed),
        'community': nx.random_partition_graph([n1, n2, n3, n - n1 - n2 - n3], .25, .01, seed=seed),
    }
    for name, G0 in graphs.items():
        # canonical node order (labels ascending): nx.random_partition_graph inserts its nodes out of label order, and
        # the reference's mapping breaks degree ties by insertion order - an adjacency matrix carries no such order
        G = nx.Graph()
        G.add_nodes_from(sorted(G0.nodes))
        G.add_edges_from(sorted((min(u, v), max(u, v)) for u, v in G0.edges))
        A = sp.csr_matrix(nx.to_scipy_sparse_array(G, nodelist=range(n), format='csr'))
        A.sort_indices()
        out = dict(indptr=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32))
        for layout in ('degree', 'community'):
            m = ref_u.generate_node_mapping(G, layout)
            new = np.array([m[i] for i in range(n)], dtype=np.int32)
            C = A.tocoo()
            newA = sp.coo_matrix((C.data, (new[C.row], new[C.col])), shape=C.shape).tocsr()
            newA.sort_indices()
            out['map_' + layout] = new
            out['new_indptr_' + layout] = newA.indptr.astype(np.int32)
            out['new_indices_' + layout] = newA.indices.astype(np.int32)
        save('layout_%s' % name, **out)


def gen_gconv():
    """G12: the dense-A GraphConvolution that dgnn.py's star import exposes (neural_dynamics.py:163-176)."""
    _, A = grid_operator(12, 'norm_lap')
    torch.manual_seed(21)
    gc = ref_nd.GraphConvolution(7, 5, bias=True)
    gc_nb = ref_nd.GraphConvolution(7, 5, bias=False)
    x = torch.randn(A.shape[0], 7)
    with torch.no_grad():
        save('gconv_dense', A=A, x=x, W=gc.fc.weight, b=gc.fc.bias, out=gc(x, A), W_nb=gc_nb.fc.weight, out_nb=gc_nb(x, A))


def gen_adams():
    """G13: the variable-coefficient Adams-Bashforth-Moulton solver (adams.py) on the NDCN ODEFunc, with one row per
    attempted step {t_n, attempted next_t, order, accepted, next next_t} recorded by wrapping (not editing) the
    reference's step method: loose / tight tolerances (orders 1-5, rejected steps), a dense tick grid (next_t clipped
    to every tick), max_order 2, the no_control right-hand side."""
    import torchdiffeq._impl.adams as ref_adams
    _, OM = grid_operator(20)
    OMs = ref_u.torch_sensor_to_torch_sparse_tensor(OM)
    cases = [('loose', .01, .001, torch.linspace(0., 5., 6), {}, {}),
             ('tight', 1e-5, 1e-7, torch.linspace(0., 2., 5), {}, {}),
             ('ticks', 1e-3, 1e-4, torch.linspace(0., 5., 30), {}, {}),
             ('order2', 1e-4, 1e-6, torch.linspace(0., 1., 4), {'max_order': 2, 'safety': 0.8}, {}),
             ('no_control', .01, .001, torch.linspace(0., 3., 7), {}, {'no_control': True})]
    orig = ref_adams.VariableCoefficientAdamsBashforth._adaptive_adams_step
    for name, rtol, atol, t, opts, fkw in cases:
        f, x = make_func(20, OMs, 41, **fkw)
        cf = CountingFunc(f)
        rows = []

        def wrapped(self, st, final_t):
            t_n, nt, order = float(st.prev_t[0]), min(float(st.next_t), float(final_t)), int(st.order)
            out = orig(self, st, final_t)
            rows.append((t_n, nt, order, 1.0 if float(out.prev_t[0]) != t_n else 0.0, float(out.next_t)))
            return out
        ref_adams.VariableCoefficientAdamsBashforth._adaptive_adams_step = wrapped
        try:
            with torch.no_grad():
                y = ref_ode.odeint(cf, x, t, rtol=rtol, atol=atol, method='adams', options=dict(opts) if opts else None)
        finally:
            ref_adams.VariableCoefficientAdamsBashforth._adaptive_adams_step = orig
        save('adams_%s' % name, x0=x, t=t, W=f.wt.weight, b=f.wt.bias, traj=y, rtol=rtol, atol=atol,
             steplog=np.array(rows, dtype=np.float64), nfe=cf.nfe, no_control=int(bool(fkw)),
             **{'opt_' + k: v for k, v in opts.items()}, **csr_of(OM))


if __name__ == '__main__':
    which = sys.argv[1:] or ['layout', 'gconv', 'rhs', 'fixed', 'dopri5', 'ndcn', 'truth', 'operators', 'dgnn', 'dataset', 'adjoint', 'resgcn', 'adams']
    for w in which:
        globals()['gen_' + w]()
