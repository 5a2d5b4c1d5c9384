"""Non-finite semantics of the path on a real MI355X: the reference's ReLU is F.relu (neural_dynamics.py:36), which
PROPAGATES NaN, and its dopri5 finiteness / step-size assertions (dopri5.py:100-102, misc.py:50-52) depend on that.
Every native variant of the right-hand side - fused2 (register gather), fused3 (group-record plan), the group-record
SpMM and the row SpMM carrying the no_control RHS, and the composed SpMM + Linear of other widths - is fed
  (a) a NaN in the state,
  (b) a NaN BORN INSIDE the evaluation: (+Inf) + (-Inf) in the accumulation of A X,
  (c) an overflow inside A X (finite inputs) that the Linear turns into NaN,
and compared with oracle.odefunc_rhs element class by element class (NaN / +Inf / -Inf / finite value).
One documented difference: the H = 256 fused kernels form W S on the fp16 matrix cores from error-free
two-piece splits of both operands (split16.h); an INFINITE entry of S = A X then makes its whole K row NaN (the row's
pieces come out Inf / NaN, and Inf times the low plane of W has both signs), where the reference's fp32 chain gives
+Inf / 0 after the ReLU per output column.  Both rows
are non-finite - what the solver's assertions look at - so for those kernels rows are compared as finite / non-finite and
every reference NaN must be a NaN."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from ndcn_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _same_classes(got, ref, tol=2e-4):
    got, ref = got.cpu().numpy(), ref.numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)), 'NaN positions differ: %d vs %d' % (np.isnan(got).sum(), np.isnan(ref).sum())
    assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= tol * max(1.0, np.abs(ref[fin]).max())


def _same_rows(got, ref, tol=2e-4):
    got, ref = got.cpu().numpy(), ref.numpy()
    assert np.all(np.isnan(got)[np.isnan(ref)]), 'a reference NaN came out as a number'
    bad_g, bad_r = ~np.isfinite(got).all(1), ~np.isfinite(ref).all(1)
    assert np.array_equal(bad_g, bad_r), 'rows with non-finite entries differ'
    fin = ~bad_r
    assert np.abs(got[fin] - ref[fin]).max() <= tol * max(1.0, np.abs(ref[fin]).max())


def _cases(n, H, side):
    g = torch.Generator().manual_seed(7)
    base = torch.rand(n, H, generator=g)
    i, j = 5 * side + 7, 5 * side + 8                  # two adjacent lattice nodes
    a = base.clone(); a[i, 3] = float('nan')
    b = base.clone(); b[i, 9] = float('inf'); b[j, 9] = float('inf')      # row i: 1 * Inf + (-w) * Inf = NaN
    c = base.clone(); c[i, :8] = 3e38                                      # 4 L: 4 * 3e38 overflows to +Inf inside A X
    return {'nan_in_state': a, 'inf_minus_inf': b, 'overflow': c}


def _operators(dev, side, scale):
    """(name, CsrOperator) for each native kernel family on the side x side lattice operator scale * L."""
    from ndcn_amd import CsrOperator, graphs
    m = (graphs.normalized_laplacian(graphs.grid_8_neighbor(side)) * scale).tocsr().astype(np.float32)
    m.sort_indices()
    plain = CsrOperator.from_scipy(m, dev)
    plain._plans_tried = True                          # no plan: fused2 / the row SpMM
    rec = CsrOperator.from_scipy(m, dev)
    rec._plans_tried = True
    rec.group_order = torch.as_tensor(rec.detect_stencil_order(), dtype=torch.int32).to(dev)
    rec.build_rec_plan(16, 40, 2)                      # fused3 / the group-record SpMM
    return m, {'no_plan': plain, 'rec_plan': rec}


@pytest.mark.parametrize('no_control', [False, True])
@pytest.mark.parametrize('H', [256, 64])
def test_rhs_propagates_non_finite_values_like_the_reference(dev, H, no_control):
    from ndcn_amd import hip, _lib
    side = 40
    n = side * side
    m, ops = _operators(dev, side, 4.0)
    A_ref = orc.coo_from_csr(m.indptr, m.indices, m.data, m.shape)
    g = torch.Generator().manual_seed(1)
    W = ((torch.rand(H, H, generator=g) - 0.5) / 8)
    b = ((torch.rand(H, generator=g) - 0.5) / 8)
    for case, X in _cases(n, H, side).items():
        ref = orc.odefunc_rhs(A_ref, X, W, b, no_control=no_control)
        assert not bool(torch.isfinite(ref).all()), case
        for name, A in ops.items():
            got = hip.rhs(A, X.to(dev), W.to(dev), b.to(dev), no_control=no_control)
            if H == 256 and not no_control:
                want = _lib.PATH_FUSED3 if name == 'rec_plan' else _lib.PATH_FUSED2
                assert _lib.load().ndcn_debug_last_rhs_path() == want
            same = _same_rows if (H == 256 and not no_control) else _same_classes
            same(got, ref)
            # the same evaluation with the stage algebra in its epilogue: K as above, y_next = y0 + c0 k0 + c1 K non-finite
            # exactly where the reference's op chain is, the error record NaN / counting the bad state entries
            y0 = torch.rand(n, H, generator=g)
            k0 = torch.randn(n, H, generator=g)
            cs = [np.float32(0.3), np.float32(-0.2)]
            K, yn = hip.rhs_rk(A, X.to(dev), W.to(dev), b.to(dev), 'combine', y0.to(dev), [k0.to(dev)], cs, no_control=no_control)
            same(K, ref)
            _same_classes(yn, y0 + (cs[0] * k0 + cs[1] * K.cpu()))
            K, (s, bad) = hip.rhs_rk(A, X.to(dev), W.to(dev), b.to(dev), 'error', y0.to(dev), [k0.to(dev)], cs, rtol=1e-2, atol=1e-3,
                                     no_control=no_control)
            same(K, ref)
            assert not np.isfinite(s)                                     # the reference's mean error ratio is NaN / Inf
            assert bad == float((~torch.isfinite(X)).sum())               # _is_finite(y1) of the NEXT step (dopri5.py:101-102)


@pytest.mark.parametrize('method', ['dopri5', 'rk4'])
@pytest.mark.parametrize('H', [256, 20])
def test_nan_born_inside_the_rhs_reaches_the_solver(dev, method, H):
    """Finite y0, but A X overflows and the Linear turns the Inf into NaN: the reference's dopri5 stops with an
    AssertionError (initial step NaN -> 'underflow in dt nan', dopri5.py:100); a fixed grid just carries the NaN on."""
    from ndcn_amd import graphs
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    side = 40
    n = side * side
    m = (graphs.normalized_laplacian(graphs.grid_8_neighbor(side)) * 4.0).tocsr().astype(np.float32)
    torch.manual_seed(0)
    f = ODEFunc(H, graphs.to_device(m, dev)).to(dev).eval()
    x0 = torch.rand(n, H)
    x0[77, :8] = 3e38
    A_ref = orc.coo_from_csr(m.indptr, m.indices, m.data, m.shape)
    fo = orc.OracleODEFunc(A_ref, f.wt.weight.detach().cpu(), f.wt.bias.detach().cpu())
    assert bool(torch.isnan(fo(0., x0)).any()) and bool(torch.isfinite(x0).all())
    t = torch.linspace(0., 1., 3)
    if method == 'dopri5':
        with pytest.raises(AssertionError):
            orc.odeint(fo, x0, t, rtol=1e-2, atol=1e-3, method='dopri5')
        with pytest.raises(AssertionError), torch.no_grad():
            ode.odeint(f, x0.to(dev), t.to(dev), rtol=1e-2, atol=1e-3, method='dopri5')
    else:
        ref = orc.odeint(fo, x0, t, method='rk4')
        with torch.no_grad():
            got = ode.odeint(f, x0.to(dev), t.to(dev), method='rk4')
        # the NaN front spreads one lattice ring per evaluation, identically on both sides
        assert np.array_equal(np.isnan(got.cpu().numpy()).any(-1), np.isnan(ref.numpy()).any(-1))
        assert bool(torch.isnan(got[-1]).any())
