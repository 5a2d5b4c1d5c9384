"""The H = 256 dense product of the fused right-hand sides (two fp16 pieces per operand, split16.h) on operands that LEAVE the
well-conditioned regime: one dominant weight, one dominant output row, log-normal / log-uniform weights, an S row with one channel
2^12 above the rest against small weights, a reference-style checkpoint through load_state_dict.  The reference's Linear is a plain
fp32 nn.Linear (neural_dynamics.py:33): a drop-in that loads its state_dicts must not lose bits because of ONE outlier.

Bound asserted per output (the one _sampled_rhs_check of test_gpu_odeint.py states):  |got - fp64| <= 2e-6 * sum of the magnitudes
that enter the output; plus trajectory L1 < 1e-4 against the oracle over a dopri5 solve with identical accept / reject decisions.
Every kernel that consumes the split planes is driven: rhs_fused3 (lattice plan), rhs_fused2 (no plan), the column sweep's dense
stage (rhs_fused3 on the identity operator), linear_gs_256_split (the VJP's gS = gZ W)."""
import numpy as np
import pytest
import torch

from oracle import ndcn_oracle as orc

pytestmark = pytest.mark.gpu
H = 256


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def split_product_only(request):
    """The tests of this file hold the SPLIT product to its stated guarantee: the range guard (round 6: weights whose rows span more than
    2^19 leave for the fp32 matrix cores, NDCN_PATH_EXACT32) is switched off around them - except where a test asks for it."""
    from ndcn_amd import _lib
    from ndcn_amd.ops import invalidate_packed_weights
    want = 1 if request.node.get_closest_marker('range_guard') else 0
    prev = _lib.load().ndcn_set_range_guard(want)
    invalidate_packed_weights()
    yield
    _lib.load().ndcn_set_range_guard(prev)
    invalidate_packed_weights()


def _default_init(seed=0):
    torch.manual_seed(seed)
    lin = torch.nn.Linear(H, H)
    return lin.weight.detach().clone(), lin.bias.detach().clone()


def _weights(kind):
    """(W, b, X column scale): the operand families of the round-4 review"""
    W, b = _default_init()
    g = torch.Generator().manual_seed(11)
    xs = None
    if kind == 'default':
        pass
    elif kind.startswith('one_weight_2^'):
        W[17, 33] *= 2.0 ** int(kind.split('^')[1])
    elif kind.startswith('one_row_2^'):
        W[17, :] *= 2.0 ** int(kind.split('^')[1])
    elif kind == 'lognormal':
        W = torch.exp(torch.randn(H, H, generator=g) * np.log(10.0)) * 1e-3 * torch.sign(torch.rand(H, H, generator=g) - 0.5)
    elif kind == 'loguniform_6_decades':
        W = 10.0 ** (torch.rand(H, H, generator=g) * 6 - 6) * torch.sign(torch.rand(H, H, generator=g) - 0.5)
    elif kind == 'channel_2^12_small_weights':                      # S channel 5 is 2^12 above the rest, its weights 2^-12 below
        xs = torch.ones(H)
        xs[5] = 2.0 ** 12
        W[:, 5] *= 2.0 ** -12
    elif kind == 'channel_2^12':
        xs = torch.ones(H)
        xs[5] = 2.0 ** 12
    else:
        raise KeyError(kind)
    return W.contiguous(), b, xs


KINDS = ['default', 'one_weight_2^8', 'one_weight_2^12', 'one_weight_2^16', 'one_row_2^12', 'lognormal', 'loguniform_6_decades',
         'channel_2^12_small_weights', 'channel_2^12']


def _operator(which, dev):
    from ndcn_amd import graphs
    if which == 'fused3':
        L = graphs.normalized_laplacian(graphs.grid_8_neighbor(48))
    elif which == 'fused2':
        L = graphs.normalized_laplacian(graphs.make_graph('power_law', 3000, seed=3))
    else:                                                            # the column sweep: G(n,p), n_cols >= 8192
        L = graphs.normalized_laplacian(graphs.make_graph('random', 9000, seed=1))
    return L, graphs.to_device(L, dev)


def _want_path(which):
    from ndcn_amd import _lib
    return {'fused3': _lib.PATH_FUSED3, 'fused2': _lib.PATH_FUSED2, 'sweep': _lib.PATH_FUSED3 | _lib.PATH_SWEEP}[which]


def _rhs_bound_check(L, A, W, b, X, dev, want_path):
    from ndcn_amd import hip, _lib
    got = hip.rhs(A, X.to(dev), W.to(dev), b.to(dev)).cpu().double().numpy()
    path = int(_lib.load().ndcn_debug_last_rhs_path())
    assert path & ~(_lib.PATH_HUB) == want_path, path
    Xh = X.numpy()
    S = orc.spmm_f64(L.indptr, L.indices, L.data, Xh)
    Wd, bd = W.double().numpy(), b.double().numpy()
    ref = np.maximum(S @ Wd.T + bd, 0)
    absS = orc.spmm_f64(L.indptr, L.indices, np.abs(L.data), np.abs(Xh))
    mag = absS @ np.abs(Wd).T + np.abs(bd)
    ratio = float((np.abs(got - ref) / (mag + 1e-300)).max())
    assert (np.abs(got - ref) <= 2e-6 * mag + 1e-30).all(), ratio
    return ratio


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('which', ['fused3', 'fused2', 'sweep'])
def test_rhs_h256_outlier_operands_within_the_fp32_grade_bound(dev, which, kind):
    L, A = _operator(which, dev)
    W, b, xs = _weights(kind)
    X = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(5))
    if xs is not None:
        X = (X * xs).contiguous()
    _rhs_bound_check(L, A, W, b, X, dev, _want_path(which))


@pytest.mark.parametrize('kind', KINDS)
def test_linear_gs_256_split_outlier_operands(dev, kind):
    """gS = (g * [Y > 0]) W on the planes of W^T (linear_bwd.hip): per output |err| <= 2e-6 sum_o |gZ_o| |W_oi|."""
    from ndcn_amd import hip
    W, b, xs = _weights(kind)
    n = 1000
    gen = torch.Generator().manual_seed(6)
    g = torch.randn(n, H, generator=gen)
    if xs is not None:
        g = (g * xs).contiguous()                                   # a dominant gradient channel against a small weight ROW of W^T
        if kind == 'channel_2^12_small_weights':
            W, b, _ = _weights('default')
            W[5, :] *= 2.0 ** -12
    Y = torch.rand(n, H, generator=gen) - 0.3
    gS, _, _ = hip.linear_bwd(g.to(dev), W.to(dev), S=None, Y=Y.to(dev), need_gS=True, need_gW=False, need_gb=False)
    gZ = (g * (Y > 0)).double().numpy()
    ref = gZ @ W.double().numpy()
    mag = np.abs(gZ) @ np.abs(W.double().numpy())
    got = gS.cpu().double().numpy()
    assert (np.abs(got - ref) <= 2e-6 * mag + 1e-30).all(), float((np.abs(got - ref) / (mag + 1e-300)).max())


@pytest.mark.parametrize('kind', ['one_weight_2^8', 'one_weight_2^12', 'lognormal', 'channel_2^12_small_weights'])
@pytest.mark.parametrize('which', ['fused3', 'fused2'])
def test_dopri5_trajectory_with_outlier_checkpoint_matches_the_oracle(dev, which, kind):
    """A reference-style checkpoint (state_dict keys of neural_dynamics.py:16: wt.weight / wt.bias) with outlier weights, loaded
    through load_state_dict, solved with dopri5: trajectory L1 < 1e-4 of its scale and the oracle's accept / reject sequence."""
    from ndcn_amd import torchdiffeq as ode
    from ndcn_amd.neural_dynamics import ODEFunc
    L, A = _operator(which, dev)
    W, b, xs = _weights(kind)
    # keep the dynamics tame: the outlier stays an outlier, the spectral size of W stays O(1)
    W = W / max(1.0, float(torch.linalg.matrix_norm(W, 2)))
    f = ODEFunc(H, A).to(dev).eval()
    f.load_state_dict({'wt.weight': W, 'wt.bias': b})
    x0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(8))
    if xs is not None:
        x0 = (x0 * xs).contiguous()
    t = torch.tensor([0., 0.5, 1.0])
    log, lo = [], []
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), rtol=.01, atol=.001, method='dopri5', step_log=log)
    fo = orc.OracleODEFunc(orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape), W, b)
    ref = orc.odeint(fo, x0, t, rtol=.01, atol=.001, method='dopri5', step_log=lo)
    assert [r[2] for r in log[:-1]] == [r[2] for r in lo]
    err = (y.cpu() - ref).abs().double()
    scale = max(1.0, float(ref.abs().max()))
    assert float(err.mean()) < 1e-4 * scale, float(err.mean()) / scale
    # the oracle's own fp32 chain is no closer to fp64 than 1e-6 of the magnitudes per evaluation: a max-abs bound at 1e-3 of scale
    assert float(err.max()) < 1e-3 * scale, float(err.max()) / scale


def test_weight_row_scales_are_powers_of_two_per_output_row(dev):
    """The pack kernel's tail: one unscale factor per output row, a power of two with max |W[n]| / unscale in [2^14, 2^15)."""
    from ndcn_amd import _lib
    from ndcn_amd.ops import ptr, stream_ptr
    lib = _lib.load()
    W, b, _ = _weights('one_row_2^12')
    W[40, :] = 0.0                                                   # an all-zero row keeps a finite scale
    Wd = W.to(dev)
    wbytes = int(lib.ndcn_rhs_work_bytes(64, H, _lib.F_RELU))
    assert wbytes >= H * H * 4 + 2 * H * H * 2 + H * 4
    work = torch.zeros(wbytes, dtype=torch.uint8, device=dev)
    from ndcn_amd import graphs, hip
    L = graphs.normalized_laplacian(graphs.grid_8_neighbor(16))
    A = graphs.to_device(L, dev)
    A.ensure_plans(H)
    X = torch.rand(256, H, device=dev)
    Y = torch.empty_like(X)
    with torch.cuda.device(dev):
        rc = lib.ndcn_rhs_f32(A.view_ref(), ptr(X), None, 256, ptr(Wd), ptr(b.to(dev)), ptr(Y), ptr(work), H, _lib.F_RELU, stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    tail = work[H * H * 4 + 2 * H * H * 2:][:H * 4].view(torch.float32).cpu().numpy().astype(np.float64)
    mant, _ = np.frexp(tail)
    assert (mant == 0.5).all()                                       # exact powers of two
    scaled = W.abs().max(dim=1).values.double().numpy() / tail
    nz = scaled > 0
    assert nz.sum() == H - 1 and (scaled[nz] >= 2.0 ** 14).all() and (scaled[nz] < 2.0 ** 15).all()


# ---- the range guard (round 6) ------------------------------------------------------------------------------------------------------

def _wide_range_weights():
    """default init, except output row 17: one weight of 1.0 against a row 2^-24 below it - and S's channel under that weight is zero,
    so that the small weights ALONE carry output 17 (the split product would keep 38 - 24 = 14 bits of them: 6e-5 of the magnitudes)"""
    W, b = _default_init()
    W[17, :] *= 2.0 ** -24 * 16.0
    W[17, 33] = 1.0
    b[17] = 0.0
    return W.contiguous(), b


@pytest.mark.range_guard
@pytest.mark.parametrize('which', ['fused3', 'fused2', 'sweep'])
def test_weights_beyond_the_guarantee_take_the_fp32_matrix_cores(dev, which):
    """nn.Linear is fp32 in the reference (neural_dynamics.py:33).  A weight row spanning 2^24 is found when the image is packed and the
    launch goes to the fp32 MFMA kernel: within 2e-6 sum |s w| of fp64 where the split product is not; ordinary weights stay on the
    split product (same path bits as ever)."""
    from ndcn_amd import hip, _lib
    L, A = _operator(which, dev)
    W, b = _wide_range_weights()
    X = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(5))
    X[:, 33] = 0.0                                                   # S[:, 33] = 0: the dominant weight meets zeros
    ratio = _rhs_bound_check(L, A, W, b, X, dev, _lib.PATH_EXACT32 | (_lib.PATH_SWEEP if which == 'sweep' else 0))   # (the sweep still forms S)
    # the same operands on the split product: outside the bound (the reason the guard exists)
    _lib.load().ndcn_set_range_guard(0)
    from ndcn_amd.ops import invalidate_packed_weights
    invalidate_packed_weights()
    with pytest.raises(AssertionError):
        _rhs_bound_check(L, A, W, b, X, dev, _want_path(which))
    _lib.load().ndcn_set_range_guard(1)
    invalidate_packed_weights()
    # ordinary weights: the fused kernels, as before
    W0, b0 = _default_init(seed=3)
    _rhs_bound_check(L, A, W0, b0, X, dev, _want_path(which))
    # a launch with an RK epilogue composes: K the same bits as the plain launch, y_next = y0 + c K
    Wd, bd, Xd = W.to(dev), b.to(dev), X.to(dev)
    y0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(6)).to(dev)
    K0 = hip.rhs(A, Xd, Wd, bd)
    K, yn = hip.rhs_rk(A, Xd, Wd, bd, 'combine', y0, [], [0.25])
    assert int(_lib.load().ndcn_debug_last_rhs_path()) & _lib.PATH_EXACT32
    assert torch.equal(K, K0)
    assert float((yn - (y0 + K0 * np.float32(0.25))).abs().max()) == 0.0
    assert ratio < 2e-6
    # the other launch modes of the fp32 route against the composed kernels: an RK4 stage, the dopri5 error record (two earlier stages)
    k1 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(7)).to(dev)
    K4, y4 = hip.rhs_rk(A, Xd, Wd, bd, 'rk4', y0, [k1], [0.125])
    assert int(_lib.load().ndcn_debug_last_rhs_path()) & _lib.PATH_EXACT32 and torch.equal(K4, K0)
    assert torch.equal(y4, hip.fixed_stage(3, y0, k1, K0, dt=0.125))
    Ke, (se, bade) = hip.rhs_rk(A, Xd, Wd, bd, 'error', y0, [k1], [0.03, -0.02], rtol=1e-2, atol=1e-3)
    sr, badr = hip.error(y0, Xd, [k1, K0], [0.03, -0.02], 1e-2, 1e-3)
    assert torch.equal(Ke, K0) and bade == badr == 0 and abs(se - sr) <= 1e-9 * abs(sr)
    if which == 'fused2':
        # ... and with a halo panel: the first 2000 rows own, the rest of X arrives as the second panel
        from ndcn_amd import CsrOperator
        sub = L[:2000]
        Ah = CsrOperator.from_scipy(sub, dev)
        Ah.lattice_hint = (0, 2000)
        got = hip.rhs(Ah, Xd[:2000].contiguous(), Wd, bd, X_halo=Xd[2000:].contiguous())
        assert int(_lib.load().ndcn_debug_last_rhs_path()) & _lib.PATH_EXACT32
        assert torch.equal(got, K0[:2000])


@pytest.mark.range_guard
@pytest.mark.parametrize('method', ['dopri5', 'rk4'])
def test_solver_with_weights_beyond_the_guarantee_matches_the_oracle(dev, method):
    """the device-resident solver learns the verdict when it packs (ndcn_solver_begin) and steps on the fp32 route: the oracle's
    trajectory and accept / reject sequence"""
    from ndcn_amd import torchdiffeq as ode, _lib
    from ndcn_amd.neural_dynamics import ODEFunc
    L, A = _operator('fused3', dev)
    W, b = _wide_range_weights()
    W = W / max(1.0, float(torch.linalg.matrix_norm(W, 2)))
    f = ODEFunc(H, A).to(dev).eval()
    f.load_state_dict({'wt.weight': W, 'wt.bias': b})
    x0 = torch.rand(L.shape[0], H, generator=torch.Generator().manual_seed(8))
    x0[:, 33] = 0.0
    t = torch.tensor([0., 0.5, 1.0])
    log, lo = [], []
    kw = dict(rtol=.01, atol=.001) if method == 'dopri5' else {}
    with torch.no_grad():
        y = ode.odeint(f, x0.to(dev), t.to(dev), method=method, **(dict(kw, step_log=log) if method == 'dopri5' else {}))
    assert int(_lib.load().ndcn_debug_last_rhs_path()) & _lib.PATH_EXACT32
    fo = orc.OracleODEFunc(orc.coo_from_csr(L.indptr, L.indices, L.data, L.shape), W, b)
    ref = orc.odeint(fo, x0, t, method=method, **(dict(kw, step_log=lo) if method == 'dopri5' else {}))
    if method == 'dopri5':
        assert [r[2] for r in log[:-1]] == [r[2] for r in lo]
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.cpu() - ref).abs().mean()) < 1e-5 * scale


@pytest.mark.range_guard
def test_backward_product_with_a_weight_column_beyond_the_guarantee(dev):
    """gS = gZ W reads the planes of W^T: its rows are W's COLUMNS.  A column spanning 2^24 whose dominant weight meets a zero of gZ is
    outside the split product's guarantee; the guard (the same scan, on the W^T image) sends the launch to the fp32 matrix cores."""
    from ndcn_amd import hip, _lib
    from ndcn_amd.ops import invalidate_packed_weights
    W, b = _default_init()
    W[:, 33] *= 2.0 ** -24 * 16.0
    W[17, 33] = 1.0
    n = 1000
    gen = torch.Generator().manual_seed(6)
    g = torch.randn(n, H, generator=gen)
    g[:, 17] = 0.0                                                   # the dominant weight of column 33 meets zeros
    Y = torch.rand(n, H, generator=gen) - 0.3
    gZ = (g * (Y > 0)).double().numpy()
    ref = gZ @ W.double().numpy()
    mag = np.abs(gZ) @ np.abs(W.double().numpy())

    def worst():
        gS, _, _ = hip.linear_bwd(g.to(dev), W.to(dev), S=None, Y=Y.to(dev), need_gS=True, need_gW=False, need_gb=False)
        return float((np.abs(gS.cpu().double().numpy() - ref) / (mag + 1e-300)).max())

    assert worst() <= 2e-6
    _lib.load().ndcn_set_range_guard(0)
    invalidate_packed_weights()
    assert worst() > 2e-6                                            # the split product on the same operands (why the guard exists)
    _lib.load().ndcn_set_range_guard(1)
    invalidate_packed_weights()
