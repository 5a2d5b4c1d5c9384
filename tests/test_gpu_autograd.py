"""Backward of the path on a real MI355X: gradients of the HIP ops / solver against
torch autograd through the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize('name', ['rhs_grid400_H20_default_coo', 'rhs_grid400_H20_no_control_coo',
                                  'rhs_grid400_H20_no_graph_coo', 'rhs_grid400_H256_default_coo'])
def test_rhs_gradients(dev, name):
    from ndcn_amd import CsrOperator
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden(name)
    H = d['W'].shape[0]
    kw = dict(no_graph='no_graph' in name, no_control='no_control' in name)
    f = ODEFunc(H, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev), **kw).to(dev)
    f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
    x = T(d['x']).to(dev).requires_grad_(True)
    g = torch.randn(400, H, generator=torch.Generator().manual_seed(0))
    y = f(torch.tensor(0.0), x)
    assert np.abs(y.detach().cpu().numpy() - d['out']).max() < 2e-5
    (y * g.to(dev)).sum().backward()
    # oracle autograd
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    W, b, xc = T(d['W']).requires_grad_(True), T(d['b']).requires_grad_(True), T(d['x']).requires_grad_(True)
    (orc.odefunc_rhs(A, xc, W, b, **kw) * g).sum().backward()
    assert rel(x.grad.cpu(), xc.grad) < 1e-4
    if not kw['no_control']:
        assert rel(f.wt.weight.grad.cpu(), W.grad) < 1e-4
        assert rel(f.wt.bias.grad.cpu(), b.grad) < 1e-4
    else:
        assert f.wt.weight.grad is None or float(f.wt.weight.grad.abs().max()) == 0.0


@pytest.mark.parametrize('method,rtol,atol,tol', [('euler', 0, 0, 2e-4), ('midpoint', 0, 0, 2e-4), ('rk4', 0, 0, 2e-4),
                                                  ('dopri5', 1e-6, 1e-8, 2e-3), ('dopri5', 1e-3, 1e-4, 2e-3),
                                                  ('dopri5', 1e-1, 1e-1, 2e-3)])
def test_backprop_through_solver(dev, method, rtol, atol, tol):
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden('fixed_rk4_equal')
    t = torch.linspace(0., 1., 5)
    f = ODEFunc(20, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)).to(dev)
    f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
    x = T(d['x0']).to(dev).requires_grad_(True)
    target = torch.rand(5, 400, 20, generator=torch.Generator().manual_seed(1))
    y = ode.odeint(f, x, t.to(dev), rtol=rtol or 1e-7, atol=atol or 1e-9, method=method)
    loss = torch.nn.functional.l1_loss(y, target.to(dev))
    loss.backward()
    A = orc.coo_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    W, b, xc = T(d['W']).requires_grad_(True), T(d['b']).requires_grad_(True), T(d['x0']).requires_grad_(True)
    fo = lambda tt, xx: orc.odefunc_rhs(A, xx, W, b)
    yo = orc.odeint(fo, xc, t, rtol=rtol or 1e-7, atol=atol or 1e-9, method=method)
    lo = torch.nn.functional.l1_loss(yo, target)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5
    # dopri5: the reference differentiates THROUGH the step-size controller (dt, t0/t1, initial step and the
    # interpolation abscissa are tensors with history, misc.py:84-170); autograd_path.py keeps those paths, so the
    # gradient must equal the oracle's full autograd - not merely the frozen-grid one (3 % away at rtol 1e-6,
    # 35 % at 1e-3, measured).
    assert rel(x.grad.cpu(), xc.grad) < tol
    assert rel(f.wt.weight.grad.cpu(), W.grad) < tol
    assert rel(f.wt.bias.grad.cpu(), b.grad) < tol


def test_ndcn_training_step_matches_reference_semantics(dev):
    """One Adam step of the heat driver's loop (heat_dynamics.py:313-334) on the HIP path vs the oracle."""
    from ndcn_amd.neural_dynamics import NDCN
    d = load_golden('ndcn_ndcn_euler')
    A = orc.dense_from_csr(d['indptr'], d['indices'], d['data'], d['shape'])
    sd = {k[4:].replace('__', '.'): T(v) for k, v in d.items() if k.startswith('sd__')}
    m = NDCN(1, 20, A.to(dev), 1, method='euler').to(dev)
    m.load_state_dict(sd)
    t, x0 = T(d['t']), T(d['x0'])
    target = T(d['out']) * 0.9 + 0.1
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-3)
    opt.zero_grad()
    pred = m(t.to(dev), x0.to(dev))
    loss = torch.nn.functional.l1_loss(pred, target.to(dev))
    loss.backward()
    # oracle
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    po = orc.ndcn_forward(ps, A, t, x0, 'euler')
    lo = torch.nn.functional.l1_loss(po, target)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-5
    for k, p in m.named_parameters():
        assert rel(p.grad.cpu(), ps[k].grad) < 5e-4, k
    # the same optimizer step on both sides lands on the same loss
    opt.step()
    names = [k for k, _ in m.named_parameters()]
    opt_o = torch.optim.Adam([ps[k] for k in names], lr=0.01, weight_decay=1e-3)
    opt_o.step()
    with torch.no_grad():
        l2 = torch.nn.functional.l1_loss(m(t.to(dev), x0.to(dev)), target.to(dev))
        l2o = torch.nn.functional.l1_loss(orc.ndcn_forward({k: v.detach() for k, v in ps.items()}, A, t, x0, 'euler'), target)
    assert abs(float(l2) - float(l2o)) < 1e-3 * max(1.0, float(l2o))


def test_forward_only_kernels_are_not_used_silently(dev):
    # requires_grad inputs must yield a graph, never a constant
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden('fixed_rk4_equal')
    f = ODEFunc(20, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)).to(dev)
    y = ode.odeint(f, T(d['x0']).to(dev), torch.tensor([0., .1, .2]).to(dev), method='dopri5', rtol=1e-3, atol=1e-4)
    assert y.requires_grad and y.grad_fn is not None


@pytest.mark.parametrize('method', ['dopri5', 'rk4'])
def test_odeint_adjoint_against_reference_gradients(dev, method):
    """odeint_adjoint (O(1)-memory backward) vs the gradients the REFERENCE's odeint_adjoint produced on the same
    inputs (fixture adjoint_*.npz, tools/gen_golden.py G9)."""
    from ndcn_amd import CsrOperator
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    d = load_golden('adjoint_' + method)
    f = ODEFunc(8, CsrOperator.from_arrays(d['indptr'], d['indices'], d['data'], d['shape'], dev)).to(dev)
    f.load_state_dict({'wt.weight': T(d['W']), 'wt.bias': T(d['b'])})
    x0 = T(d['x0']).to(dev).requires_grad_(True)
    kw = dict(method=method) if method == 'rk4' else dict(method=method, rtol=float(d['rtol']), atol=float(d['atol']))
    y = ode.odeint_adjoint(f, x0, T(d['t']).to(dev), **kw)
    assert np.abs(y.detach().cpu().numpy() - d['traj']).max() < 1e-5
    loss = torch.nn.functional.l1_loss(y, T(d['target']).to(dev))
    assert abs(float(loss.detach()) - float(d['loss'])) < 1e-6
    loss.backward()
    # rk4: the backward pass is deterministic given the grid; dopri5: the backward solve chooses its own steps at
    # rtol 1e-5, so the two implementations agree to the solver tolerance, not to rounding (5e-3 measured)
    tol = 2e-3 if method == 'rk4' else 1e-2
    assert rel(x0.grad.cpu(), T(d['g_x0'])) < tol
    assert rel(f.wt.weight.grad.cpu(), T(d['g_W'])) < tol
    assert rel(f.wt.bias.grad.cpu(), T(d['g_b'])) < tol
    # ODEBlock(adjoint=True) routes here (neural_dynamics.py:72-74)
    from ndcn_amd.neural_dynamics import ODEBlock
    blk = ODEBlock(f, rtol=1e-3, atol=1e-4, method='dopri5', adjoint=True, terminal=True)
    out = blk(T(d['t']).to(dev), T(d['x0']).to(dev))
    assert out.requires_grad and out.shape == (144, 8)


@pytest.mark.parametrize('n,Hi,Ho', [(400, 20, 20), (1000, 256, 256), (777, 1, 20), (777, 20, 1), (3001, 64, 16), (130, 300, 40),
                                     (9, 256, 256), (70000, 256, 256)])
@pytest.mark.parametrize('masked', [False, True])
def test_linear_backward_kernels(dev, n, Hi, Ho, masked):
    """ndcn_linear_bwd_f32 (gS = gZ W as an MFMA GEMM with W read transposed, gW = gZ^T S split over row chunks with a
    fixed-order sum, gb with it, ReLU mask fused into the operand loads) against fp64; deterministic run to run."""
    from ndcn_amd import hip
    gen = torch.Generator().manual_seed(n + Hi)
    S = torch.randn(n, Hi, generator=gen).to(dev)
    W = (torch.randn(Ho, Hi, generator=gen) / 4).to(dev)
    g = torch.randn(n, Ho, generator=gen).to(dev)
    Y = torch.relu(torch.randn(n, Ho, generator=gen)).to(dev) if masked else None
    gS, gW, gb = hip.linear_bwd(g, W, S=S, Y=Y)
    gZ = (g * (Y > 0)).double() if masked else g.double()
    scale = float(np.sqrt(n))
    assert float((gS.double() - gZ @ W.double()).abs().max()) < 2e-5 * max(1.0, Ho ** 0.5)
    assert float((gW.double() - gZ.t() @ S.double()).abs().max()) < 3e-5 * scale * 4
    assert float((gb.double() - gZ.sum(0)).abs().max()) < 3e-5 * scale * 4
    gS2, gW2, gb2 = hip.linear_bwd(g, W, S=S, Y=Y)
    assert torch.equal(gW, gW2) and torch.equal(gb, gb2) and torch.equal(gS, gS2)
    only = hip.linear_bwd(g, W, S=S, Y=Y, need_gS=False, need_gb=False)
    assert only[0] is None and only[2] is None and torch.equal(only[1], gW)
    if masked:
        assert torch.equal(hip.relu_bwd(g, Y), g * (Y > 0))
    assert torch.equal(hip.scale(g, -0.37), g * np.float32(-0.37))
