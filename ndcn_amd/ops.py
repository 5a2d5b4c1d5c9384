"""torch-tensor front end of the C-ABI kernels (include/ndcn_hip.h).

PyTorch supplies device memory and the current HIP stream; every computation below is a call into
libndcn_hip.so.  Host tensors are refused (`_lib.require_device`) - there is no CPU path in the product.

`HipOps` is the one shipped implementation of the small "panel ops" interface the integrator's host
logic is written against (ndcn_amd/torchdiffeq/_impl/core.py).  Tests drive that same host logic with
an oracle-backed double to check the control flow without a GPU; product code never does.
"""
import ctypes
import os
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_device, stream_ptr
from .csr import CsrOperator, as_csr

_F = ctypes.c_float
_P = ctypes.c_void_p


def _panel(t, what='panel'):
    require_device(t, what)
    return t if t.is_contiguous() else t.contiguous()


def _terms(ks, cs):
    n = len(ks)
    arr_k = (_P * n)(*[k.data_ptr() for k in ks])
    arr_c = (_F * n)(*[float(c) for c in cs])
    return arr_k, arr_c, n


# The reductions below share per-device scratch (partials, the result record, its pinned host mirror).  Under autograd
# they are reached from TWO threads at once - the engine runs the backward of nodes whose gradient lives on the GPU on its
# device thread and the nodes of the solver's CPU scalar chain (error ratio, norms) on the calling thread - so launch +
# read-back of one reduction is one critical section.
_REDUCE_LOCK = threading.RLock()


class _Reducer:
    """Per-device scratch for the two-pass reductions + a pinned host mirror for the 16-byte record."""
    _by_device = {}

    @classmethod
    def get(cls, device):
        r = cls._by_device.get(device)
        if r is None:
            r = cls()
            lib = _lib.load()
            r.ws = torch.empty(int(lib.ndcn_reduce_ws_bytes()), dtype=torch.uint8, device=device)
            r.out = torch.zeros(2, dtype=torch.float64, device=device)
            r.host = torch.zeros(2, dtype=torch.float64).pin_memory()
            cls._by_device[device] = r
        return r

    def fetch(self):
        self.host.copy_(self.out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return float(self.host[0]), float(self.host[1])


class ErrorRecord:
    """A caller-owned {sum of squared error ratios, non-finite count} record: two doubles on the device + a pinned host
    mirror.  An error evaluation that is split into several launches (row blocks of a shard: NDCN_F_ACCUM) accumulates
    into ITS OWN record - the per-device one of _Reducer is shared with every other reduction of the process, and the
    lock is released between the launches of a split."""

    def __init__(self, device):
        self.out = torch.zeros(2, dtype=torch.float64, device=device)
        self.host = torch.zeros(2, dtype=torch.float64).pin_memory()

    fetch = _Reducer.fetch


class _PackedWeights:
    """The fused H = 256 kernels read W in a packed, split form that ndcn_rhs_f32 / ndcn_rhs_rk_f32 build in their
    scratch at every call (two small launches in front of the big one).  A solver calls them thousands of times with
    the same weights: the scratch is kept per weight TENSOR OBJECT (weak reference: a new tensor that happens to reuse the
    address or the id of a dead one never hits), version counter (in-place updates - optimizers, load_state_dict - move
    it), stream and EPOCH; later calls pass NDCN_F_PACKED.

    Epoch (round 5): `W.data.add_()`, `W.data.copy_()`, `dist.broadcast(W.data)` and raw-pointer writes bypass the version
    counter - hand-written SGD / EMA / clipping code does exactly that between two solves.  Every `odeint` / `odeint_adjoint`
    call therefore starts a new epoch (`new_epoch`): the first use of a weight inside it re-packs (two small launches, ~10 us
    per solve), the thousands that follow hit.  What is left to the caller: such a write BETWEEN two direct `hip.rhs` calls
    outside any solve (or in the middle of one - which no autograd graph survives either): call
    `ndcn_amd.ops.invalidate_packed_weights()` after it, or run with NDCN_PACK_CACHE=0 (re-pack at every call)."""
    _cache = {}
    _epoch = 0
    enabled = os.environ.get('NDCN_PACK_CACHE', '1') != '0'

    @classmethod
    def invalidate(cls):
        cls._cache.clear()

    @classmethod
    def new_epoch(cls):
        cls._epoch += 1

    @classmethod
    def get(cls, W, nbytes, tag='fwd'):
        if not cls.enabled:
            return torch.empty(nbytes, dtype=torch.uint8, device=W.device), 0
        # a whole-tensor alias (the training path hands W on from evaluation to evaluation as `W.view_as(W)`: a new tensor object each
        # time) is the tensor it views: same storage, same version counter
        base = W._base
        if base is not None and base.data_ptr() == W.data_ptr() and base.shape == W.shape and base.is_contiguous():
            W = base
        key = (id(W), torch.cuda.current_stream(W.device).cuda_stream, nbytes, tag)
        hit = cls._cache.get(key)
        if hit is not None and hit[0]() is W and hit[1] == W._version and hit[2] == W.data_ptr():
            if hit[4] == cls._epoch:
                return hit[3], _lib.F_PACKED
            cls._cache[key] = hit[:4] + (cls._epoch,)             # same buffer, packed again by this call (stream-ordered)
            return hit[3], 0
        if len(cls._cache) >= 8:
            dead = [k for k, v in cls._cache.items() if v[0]() is None]
            for k in dead or [next(iter(cls._cache))]:
                cls._cache.pop(k)
        work = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
        cls._cache[key] = (weakref.ref(W), W._version, W.data_ptr(), work, cls._epoch)
        return work, 0


class _BwdDots:
    """Per-device scratch of the solver VJP kernels: partial sums + the 8-double result + its pinned host mirror."""
    _by_device = {}

    @classmethod
    def get(cls, device):
        r = cls._by_device.get(device)
        if r is None:
            r = cls()
            r.ws = torch.empty(int(_lib.load().ndcn_rk_bwd_ws_bytes()), dtype=torch.uint8, device=device)
            r.out = torch.zeros(8, dtype=torch.float64, device=device)
            r.host = torch.zeros(8, dtype=torch.float64).pin_memory()
            cls._by_device[device] = r
        return r

    def fetch(self):
        self.host.copy_(self.out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.host.tolist()

    # ---- deferred read-back ------------------------------------------------------------------------------------------
    # A training step's backward makes ~35 of these reductions, and `fetch` stops the host at each of them until the GPU has
    # drained - with nothing queued behind: 6.7 ms of idle GPU per step on the 100k-node case (rocprofv3 kernel trace, 17 %).
    # `lazy` instead returns 0-d float32 HOST tensors (views of a pinned ring) that an asynchronous copy fills, plus an event;
    # whoever reads them first calls `await_lazy` (autograd_path._Await, the only consumer): by then the panel launches of the
    # whole solver step are queued behind the copy.  Same bits as `fetch`: the fp64 sums, times `scale` in fp64, rounded to fp32.
    RING = 1 << 15
    _ring_lock = threading.Lock()
    _ring = None
    _ring_pos = 0
    _pending = {}

    def lazy(self, n, scale=1.0):
        cls = _BwdDots
        src = self.out[:n]
        m = int(n)
        with cls._ring_lock:                # (ring position and pending table: written by the device's autograd worker, read by the caller's thread)
            if cls._ring is None:
                cls._ring = torch.zeros(cls.RING, dtype=torch.float32).pin_memory()
            if cls._ring_pos + m > cls.RING:
                cls._ring_pos = 0
            lo = cls._ring_pos
            cls._ring_pos += m
        dst = cls._ring[lo:lo + m]
        vals = (src * scale) if scale != 1.0 else src
        dst.copy_(vals.to(torch.float32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        outs = [dst[i] for i in range(m)]
        rec = (ev, [o.data_ptr() for o in outs])
        with cls._ring_lock:
            for p in rec[1]:
                cls._pending[p] = rec      # (a slot the ring hands out again replaces the stale record of its previous use)
        return outs

    @classmethod
    def await_lazy(cls, t):
        if t is not None and cls._pending:
            with cls._ring_lock:
                rec = cls._pending.get(t.data_ptr())
            if rec is not None:
                rec[0].synchronize()
                with cls._ring_lock:
                    for p in rec[1]:       # one copy filled all the slots of this read-back: none of them needs the event again
                        if cls._pending.get(p) is rec:
                            del cls._pending[p]
        return t


def _grad_ptrs(like, needs):
    outs = [torch.empty_like(like) if n else None for n in needs]
    return outs, (_P * len(needs))(*[None if o is None else o.data_ptr() for o in outs])


def _acc_ptrs(accs, n):
    """nullable array of the gradients the terms have ALREADY received (accumulating VJPs, include/ndcn_hip.h); the tensors
    are kept alive by the caller for the duration of the launch (stream-ordered free of the caching allocator)."""
    if accs is None or all(a is None for a in accs):
        return None
    assert len(accs) == n
    return (_P * n)(*[None if a is None else _panel(a, 'received gradient').data_ptr() for a in accs])


class HipOps:
    """Panel operations on fp32 CUDA tensors, each ONE kernel launch through the C-ABI."""

    name = 'hip'

    # ---------------------------------------------------------------- graph convolution pieces
    @staticmethod
    def spmm(A, X, X_halo=None, alpha=1.0, relu=False, out=None):
        """alpha * (A X) [relu].  Replaces torch.sparse.mm / torch.mm(A, x) (neural_dynamics.py:29,31)."""
        A = as_csr(A)
        X = _panel(X)
        squeeze = X.dim() == 1
        X2 = X.view(-1, 1) if squeeze else X
        assert X2.dim() == 2
        H = X2.shape[1]
        n_own = X2.shape[0]
        if X_halo is not None:
            X_halo = _panel(X_halo, 'halo panel')
            assert X_halo.shape[1] == H
        assert A.shape[1] == n_own + (0 if X_halo is None else X_halo.shape[0]), \
            'operator has %d columns, panels supply %d rows' % (A.shape[1], n_own + (0 if X_halo is None else X_halo.shape[0]))
        if A.device != X2.device:
            raise _lib.NdcnHipError(_lib.EINVAL, 'operator on %s, panel on %s' % (A.device, X2.device))
        A.ensure_plans(H)
        Y = out if out is not None else torch.empty((A.shape[0], H), dtype=torch.float32, device=X2.device)
        with torch.cuda.device(X2.device):
            check(_lib.load().ndcn_spmm_f32(A.view_ref(), ptr(X2), ptr(X_halo), n_own, ptr(Y), H, float(alpha),
                                            _lib.F_RELU if relu else 0, stream_ptr()))
        return Y.view(-1) if squeeze else Y

    @staticmethod
    def adjoint_rhs(A, y, a, W, b, no_graph=False, no_control=False):
        """ndcn_adjoint_rhs_f32: (K, vjp_y, vjp_W, vjp_b) = (ODEFunc(y), -A^T((a.[K>0]) W), -(a.[K>0])^T (A y), -sum_rows a.[K>0]) -
        the right-hand side of odeint_adjoint's augmented system (adjoint.py:34-59) without a torch graph; vjp_W / vjp_b are
        None under no_control."""
        y, a = _panel(y), _panel(a)
        H = y.shape[1]
        flags = _lib.F_RELU | (_lib.F_NO_GRAPH if no_graph else 0) | (_lib.F_NO_CONTROL if no_control else 0)
        lib = _lib.load()
        if no_graph:
            view = view_t = ctypes.byref(_lib.empty_csr(y.shape[0]))
        else:
            A = as_csr(A)
            A.ensure_plans(H)
            At = A.transpose()
            At.ensure_plans(H)
            view, view_t = A.view_ref(), At.view_ref()
        K, vjp_y = torch.empty_like(y), torch.empty_like(y)
        vW = vb = None
        if not no_control:
            W = _panel(W, 'weight')
            b = _panel(b, 'bias') if b is not None else None
            vW = torch.empty((H, H), dtype=torch.float32, device=y.device)
            vb = torch.empty((H,), dtype=torch.float32, device=y.device)
        work = torch.empty(int(lib.ndcn_adjoint_rhs_work_bytes(y.shape[0], H, flags)) + 256, dtype=torch.uint8, device=y.device)
        off = (-work.data_ptr()) % 256
        with torch.cuda.device(y.device):
            check(lib.ndcn_adjoint_rhs_f32(view, view_t, ptr(y), ptr(a), ptr(None if no_control else W), ptr(None if no_control else b),
                                           ptr(K), ptr(vjp_y), ptr(vW), ptr(vb), ctypes.c_void_p(work.data_ptr() + off), H, flags,
                                           stream_ptr()))
        return K, vjp_y, vW, vb

    @staticmethod
    def gcn(A, X, W, b=None, relu=False):
        """A (X W^T + b) [relu]: GraphConvolution.forward of the reference (models.py:14-18) in one library call."""
        A = as_csr(A)
        X, W = _panel(X), _panel(W, 'weight')
        if b is not None:
            b = _panel(b, 'bias')
        Ho, Hi = W.shape
        assert X.dim() == 2 and X.shape == (A.shape[1], Hi)
        A.ensure_plans(Ho)
        lib = _lib.load()
        Y = torch.empty((A.shape[0], Ho), dtype=torch.float32, device=X.device)
        work = torch.empty(int(lib.ndcn_gcn_work_bytes(A.shape[1], Ho)), dtype=torch.uint8, device=X.device)
        with torch.cuda.device(X.device):
            check(lib.ndcn_gcn_f32(A.view_ref(), ptr(X), ptr(W), ptr(b), ptr(Y), ptr(work), Hi, Ho,
                                   _lib.F_RELU if relu else 0, stream_ptr()))
        return Y

    @staticmethod
    def linear(S, W, b=None, relu=False):
        """S W^T + b [relu] over the last dimension (nn.Linear semantics; neural_dynamics.py:33,143-148)."""
        S = _panel(S)
        W = _panel(W, 'weight')
        if b is not None:
            b = _panel(b, 'bias')
        lead = S.shape[:-1]
        Hi, Ho = S.shape[-1], W.shape[0]
        assert W.shape[1] == Hi
        S2 = S.reshape(-1, Hi)
        Y = torch.empty((S2.shape[0], Ho), dtype=torch.float32, device=S.device)
        with torch.cuda.device(S.device):
            check(_lib.load().ndcn_linear_f32(ptr(S2), ptr(W), ptr(b), ptr(Y), S2.shape[0], Hi, Ho,
                                              _lib.F_RELU if relu else 0, stream_ptr()))
        return Y.view(*lead, Ho)

    @staticmethod
    def linear_bwd(g, W, S=None, Y=None, need_gS=True, need_gW=True, need_gb=True):
        """Backward of act(S W^T + b): returns (gS, gW, gb) (None where not requested).  Y: the ReLU output (mask
        gZ = g where Y > 0) or None.  One GEMM launch for gS, one split-row launch + fixed-order sum for gW / gb."""
        g = _panel(g)
        W = _panel(W, 'weight')
        Ho, Hi = W.shape
        lead = g.shape[:-1]
        g2 = g.reshape(-1, Ho)
        n = g2.shape[0]
        S2 = _panel(S).reshape(-1, Hi) if S is not None else None
        Y2 = _panel(Y).reshape(-1, Ho) if Y is not None else None
        need_gW = need_gW and S2 is not None
        lib = _lib.load()
        gS = torch.empty((n, Hi), dtype=torch.float32, device=g.device) if need_gS else None
        gW = torch.empty((Ho, Hi), dtype=torch.float32, device=g.device) if need_gW else None
        gb = torch.empty((Ho,), dtype=torch.float32, device=g.device) if need_gb else None
        work, flags = None, 0
        if need_gW or need_gb or (need_gS and Hi == 256 and Ho == 256):      # (H = 256: gS packs the planes of W^T there)
            nbytes = int(lib.ndcn_linear_bwd_work_bytes(n, Hi, Ho))
            if need_gS and Hi == 256 and Ho == 256:
                # the scratch is kept per weight tensor (object, version, stream): the planes of W^T packed by the first VJP of a
                # backward pass serve the dozens that follow (the head of the buffer is per-call scratch, stream-ordered)
                work, flags = _PackedWeights.get(W, nbytes, tag='bwd')
            else:
                work = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            check(lib.ndcn_linear_bwd_f32(ptr(g2), ptr(Y2), ptr(S2 if S2 is not None else g2), ptr(W), ptr(gS), ptr(gW), ptr(gb),
                                          ptr(work), n, Hi, Ho, flags, stream_ptr()))
        return (gS.view(*lead, Hi) if gS is not None else None), gW, gb

    @staticmethod
    def relu_bwd(g, y):
        """g where y > 0 else 0 (VJP of relu from its output)."""
        g, y = _panel(g), _panel(y)
        out = torch.empty_like(g)
        with torch.cuda.device(g.device):
            check(_lib.load().ndcn_relu_bwd_f32(ptr(out), ptr(g), ptr(y), g.numel(), stream_ptr()))
        return out

    @staticmethod
    def copy(x, out=None):
        """out = x as the library's streaming pass (ndcn_copy_f32)."""
        x = _panel(x)
        if out is None:
            out = torch.empty_like(x)
        assert out.is_contiguous() and out.numel() == x.numel()
        with torch.cuda.device(x.device):
            check(_lib.load().ndcn_copy_f32(ptr(out), ptr(x), x.numel(), stream_ptr()))
        return out

    @staticmethod
    def scale(x, w):
        """w * x as one streaming kernel."""
        x = _panel(x)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(_lib.load().ndcn_scale_f32(ptr(out), ptr(x), float(w), x.numel(), stream_ptr()))
        return out

    @staticmethod
    def rhs(A, X, W, b, no_graph=False, no_control=False, X_halo=None, out=None, relu=True):
        """The whole ODEFunc.forward: relu(W (A X) + b)   (neural_dynamics.py:20-39, dropout 0).
        relu=False: the same launch without the activation - the transposed half of the adjoint right-hand side,
        (A^T gZ) W with the operator / weight pair (A^T, W^T) and no bias (_impl/adjoint_fused.py)."""
        X = _panel(X)
        H = X.shape[1]
        flags = (_lib.F_RELU if relu else 0) | (_lib.F_NO_GRAPH if no_graph else 0) | (_lib.F_NO_CONTROL if no_control else 0)
        lib = _lib.load()
        if no_graph:
            view = _lib.empty_csr(X.shape[0])
            view_ref = ctypes.byref(view)
            n_rows = X.shape[0]
        else:
            A = as_csr(A)
            if A.device != X.device:
                raise _lib.NdcnHipError(_lib.EINVAL, 'operator on %s, panel on %s' % (A.device, X.device))
            A.ensure_plans(H)
            view_ref = A.view_ref()
            n_rows = A.shape[0]
        if not no_control:
            W = _panel(W, 'weight')
            b = _panel(b, 'bias') if b is not None else None
            assert W.shape == (H, H)
        if X_halo is not None:
            X_halo = _panel(X_halo, 'halo panel')
        Y = out if out is not None else torch.empty((n_rows, H), dtype=torch.float32, device=X.device)
        work = None
        wbytes = int(lib.ndcn_rhs_work_bytes(n_rows, H, flags))
        if wbytes:
            if H == 256 and not no_control and not no_graph and not torch.is_grad_enabled():
                work, packed = _PackedWeights.get(W, wbytes)
                flags |= packed
            else:
                work = torch.empty(wbytes, dtype=torch.uint8, device=X.device)
        with torch.cuda.device(X.device):
            check(lib.ndcn_rhs_f32(view_ref, ptr(X), ptr(X_halo), X.shape[0], ptr(None if no_control else W),
                                   ptr(None if no_control else b), ptr(Y), ptr(work), H, flags, stream_ptr()))
        return Y

    @staticmethod
    def rhs_rk(A, X, W, b, mode, y0, kprev, cs, rtol=0.0, atol=0.0, no_graph=False, no_control=False, X_halo=None,
               out_K=None, out_y=None, y1=None, accum=False, fetch=True, aux_cs=None, out_aux=None, record=None, relu=True,
               x_mask=None, s_out=None):
        """K = ODEFunc(X) plus, in the same pass, the stage algebra consuming K (ndcn_rhs_rk_f32).
        record (mode 'error'): an ErrorRecord of the caller's that receives / accumulates the result instead of the
        per-device one (split evaluations: new_error_record()).
        mode 'combine': returns (K, y0 + sum cs[m] kprev[m] + cs[-1] K); mode 'error': returns
        (K, (sum of squared error ratios, non-finite count of X)) - the dopri5 error record with X = y1;
        mode 'rk4': stage len(kprev) of the 3/8-rule step, cs = [dt]: returns (K, next stage input / step result).
        mode 'error' only - y1: the rows of the state whose error record is formed (default X: the single-launch case);
        accum: add the record to the one already in the device buffer (an evaluation split into several launches);
        fetch=False: leave the record on the device (returns (K, None); the last launch of the split fetches).
        mode 'combine' only - aux_cs: coefficients of a second linear combination of the same stages (no y0), formed in
        the same pass: returns (K, y_next, sum aux_cs[m] kprev[m] + aux_cs[-1] K) - dopri5's partial error sum E.
        x_mask / s_out (ndcn_rhs_rk_adj_f32, where rhs_adj_supported says so): the input is X (.) [x_mask > 0] formed on the staged
        rows / S = A X is written into s_out too - the two halves of odeint_adjoint's right-hand side (_impl/adjoint_fused.py)."""
        X = _panel(X)
        H = X.shape[1]
        flags = (_lib.F_RELU if relu else 0) | (_lib.F_NO_GRAPH if no_graph else 0) | (_lib.F_NO_CONTROL if no_control else 0)
        lib = _lib.load()
        if no_graph:
            view_ref = ctypes.byref(_lib.empty_csr(X.shape[0]))
            n_rows = X.shape[0]
        else:
            A = as_csr(A)
            A.ensure_plans(H)
            view_ref = A.view_ref()
            n_rows = A.shape[0]
        if not no_control:
            W = _panel(W, 'weight')
            b = _panel(b, 'bias') if b is not None else None
        y0 = _panel(y0)
        kprev = [_panel(k) for k in kprev]
        assert len(cs) == (1 if mode == 'rk4' else len(kprev) + 1)
        if X_halo is not None:
            X_halo = _panel(X_halo, 'halo panel')
        if y1 is not None:
            y1 = _panel(y1, 'y1')
            assert mode == 'error' and tuple(y1.shape) == (n_rows, H)
        if accum:
            flags |= _lib.F_ACCUM
        K = out_K if out_K is not None else torch.empty((n_rows, H), dtype=torch.float32, device=X.device)
        assert K.is_contiguous() and tuple(K.shape) == (n_rows, H)
        wbytes = int(lib.ndcn_rhs_work_bytes(n_rows, H, flags))
        work = None
        if wbytes:
            if H == 256 and not no_control and not no_graph and not torch.is_grad_enabled():
                work, packed = _PackedWeights.get(W, wbytes)
                flags |= packed
            else:
                work = torch.empty(wbytes, dtype=torch.uint8, device=X.device)
        arr_k = (_P * max(len(kprev), 1))(*[k.data_ptr() for k in kprev])
        arr_c = (_F * len(cs))(*[float(c) for c in cs])
        y_aux = arr_c2 = None
        if aux_cs is not None:
            assert mode == 'combine' and len(aux_cs) == len(kprev) + 1
            arr_c2 = (_F * len(aux_cs))(*[float(c) for c in aux_cs])
            y_aux = out_aux if out_aux is not None else torch.empty_like(K)
        rk = {'combine': _lib.RK_COMBINE, 'error': _lib.RK_ERROR, 'rk4': _lib.RK_RK4}[mode]
        y_next = (out_y if out_y is not None else torch.empty_like(K)) if mode in ('combine', 'rk4') else None
        red = _Reducer.get(X.device)
        if x_mask is not None or s_out is not None:
            assert X_halo is None and aux_cs is None and not accum
            x_mask = _panel(x_mask, 'mask panel') if x_mask is not None else None
            with _REDUCE_LOCK, torch.cuda.device(X.device):
                check(lib.ndcn_rhs_rk_adj_f32(view_ref, ptr(X), ptr(x_mask), ptr(s_out), ptr(W), ptr(b), ptr(K), ptr(work), H, flags, rk,
                                              ptr(y0), arr_k, arr_c, len(kprev), ptr(y_next), ptr(y1), float(rtol), float(atol),
                                              ptr((record or red).out), ptr(red.ws), stream_ptr()))
                if mode == 'combine':
                    return K, y_next
                return K, ((record or red).fetch() if fetch else None)
        with _REDUCE_LOCK, torch.cuda.device(X.device):
            check(lib.ndcn_rhs_rk_f32(view_ref, ptr(X), ptr(X_halo), X.shape[0], ptr(None if no_control else W),
                                      ptr(None if no_control else b), ptr(K), ptr(work), H, flags, rk, ptr(y0), arr_k,
                                      arr_c, len(kprev), ptr(y_next), ptr(y1), ptr(y_aux), arr_c2, float(rtol), float(atol),
                                      ptr((record or red).out), ptr(red.ws), stream_ptr()))
            if aux_cs is not None:
                return K, y_next, y_aux
            if mode in ('combine', 'rk4'):
                return K, y_next
            return K, ((record or red).fetch() if fetch else None)

    @staticmethod
    def rhs_adj_supported(A, H, mode, n_prev, no_control=False):
        """can rhs_rk honour x_mask / s_out for this operator and launch?"""
        A = as_csr(A)
        A.ensure_plans(H)
        flags = _lib.F_NO_CONTROL if no_control else 0
        rk = {'combine': _lib.RK_COMBINE, 'error': _lib.RK_ERROR}.get(mode, 0)
        return bool(_lib.load().ndcn_rhs_adj_supported(A.view_ref(), H, flags, rk, n_prev))

    @staticmethod
    def new_error_record(device):
        return ErrorRecord(device)

    @staticmethod
    def rhs_rk_xadd(A, X, xadd, xadd_c, W, b, y0, k_prev, cs):
        """The evaluation that opens a dopri5 step (ndcn_rhs_rk_xadd_f32): K = ODEFunc(X + xadd_c * xadd) and
        y_next = y0 + (cs[0] * k_prev + cs[1] * K) in one pass, the input formed on the rows the kernel stages - the bits of
        combine(X, [xadd], [xadd_c]) followed by rhs_rk(..., 'combine', y0, [k_prev], cs).  Returns (K, y_next), or None where
        the operator has no such kernel (ndcn_rhs_xadd_supported)."""
        X, xadd, y0, k_prev = _panel(X), _panel(xadd), _panel(y0), _panel(k_prev)
        W = _panel(W, 'weight')
        b = _panel(b, 'bias') if b is not None else None
        H = X.shape[1]
        A = as_csr(A)
        A.ensure_plans(H)
        lib = _lib.load()
        flags = _lib.F_RELU
        if not int(lib.ndcn_rhs_xadd_supported(A.view_ref(), H, flags, _lib.RK_COMBINE, 1)):
            return None
        assert len(cs) == 2
        K, y_next = torch.empty_like(X), torch.empty_like(X)
        work = torch.empty(int(lib.ndcn_rhs_work_bytes(A.shape[0], H, flags)), dtype=torch.uint8, device=X.device)
        arr_c = (_F * 2)(float(cs[0]), float(cs[1]))
        with torch.cuda.device(X.device):
            check(lib.ndcn_rhs_rk_xadd_f32(A.view_ref(), ptr(X), ptr(xadd), float(xadd_c), ptr(W), ptr(b), ptr(K), ptr(work), H, flags,
                                           ptr(y0), ptr(k_prev), arr_c, ptr(y_next), stream_ptr()))
        return K, y_next

    @staticmethod
    def gather_rows(X, idx):
        X = _panel(X)
        assert idx.dtype == torch.int32 and idx.is_cuda
        out = torch.empty((idx.numel(), X.shape[1]), dtype=torch.float32, device=X.device)
        with torch.cuda.device(X.device):
            check(_lib.load().ndcn_gather_rows_f32(ptr(X), ptr(idx), idx.numel(), X.shape[1], ptr(out), stream_ptr()))
        return out

    # ---------------------------------------------------------------- Runge-Kutta bookkeeping
    @staticmethod
    def combine(y0, ks, cs):
        """y0 + sum_j cs[j] * ks[j]  (cs already dt*beta in fp32; misc.py:22-25 order and rounding)."""
        y0 = _panel(y0)
        ks = [_panel(k) for k in ks]
        out = torch.empty_like(y0)
        arr_k, arr_c, n = _terms(ks, cs)
        with torch.cuda.device(y0.device):
            check(_lib.load().ndcn_rk_combine_f32(ptr(out), ptr(y0), arr_k, arr_c, n, y0.numel(), stream_ptr()))
        return out

    @staticmethod
    def lincomb(ks, cs, y0=None):
        """[y0 +] sum_j cs[j] * ks[j] in one pass (ndcn_rk_combine_f32 with or without the unit-coefficient term)."""
        ks = [_panel(k) for k in ks]
        out = torch.empty_like(ks[0])
        arr_k, arr_c, n = _terms(ks, cs)
        with torch.cuda.device(out.device):
            check(_lib.load().ndcn_rk_combine_f32(ptr(out), ptr(_panel(y0)) if y0 is not None else None, arr_k, arr_c, n, out.numel(),
                                                  stream_ptr()))
        return out

    @staticmethod
    def error(y0, y1, ks, cs, rtol, atol):
        """(sum of squared error ratios, non-finite count of y1) as host floats; one 16-byte read-back."""
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        red = _Reducer.get(y0.device)
        arr_k, arr_c, n = _terms(ks, cs)
        with _REDUCE_LOCK, torch.cuda.device(y0.device):
            check(_lib.load().ndcn_rk_error_f32(ptr(y0), ptr(y1), arr_k, arr_c, n, float(rtol), float(atol), y0.numel(),
                                                ptr(red.out), ptr(red.ws), stream_ptr()))
            return red.fetch()

    @staticmethod
    def scaled_sumsq(a, b, y, rtol, atol):
        """sum(((a - b) / (atol + |y| rtol))^2) and the non-finite count of a  (misc.py:121-138)."""
        a, y = _panel(a), _panel(y)
        b = _panel(b) if b is not None else None
        red = _Reducer.get(a.device)
        with _REDUCE_LOCK, torch.cuda.device(a.device):
            check(_lib.load().ndcn_scaled_sumsq_f32(ptr(a), ptr(b), ptr(y), float(rtol), float(atol), a.numel(),
                                                    ptr(red.out), ptr(red.ws), stream_ptr()))
            return red.fetch()

    # ---------------------------------------------------------------- VJPs of the dopri5 panel operations
    @staticmethod
    def combine_bwd(g, ks, cs, need_k, need_dots=True, accs=None, acc_y0=None, lazy=False):
        """VJP of combine: ([c_j g or None], [<g, k_j>] as host floats or None).  accs / acc_y0: gradients the terms / y0 have
        already received - the outputs become acc_j + c_j g (and a third result acc_y0 + g is returned when acc_y0 is given).
        lazy: the inner products as deferred 0-d float32 host tensors (_BwdDots.lazy) instead of floats."""
        g = _panel(g)
        ks = [_panel(k) for k in ks]
        arr_k, arr_c, n = _terms(ks, cs)
        gk, arr_g = _grad_ptrs(g, need_k)
        arr_a = _acc_ptrs(accs, n)
        gy0 = torch.empty_like(g) if acc_y0 is not None else None
        d = _BwdDots.get(g.device)
        with _REDUCE_LOCK, torch.cuda.device(g.device):
            check(_lib.load().ndcn_rk_combine_bwd_f32(ptr(g), arr_k, arr_c, n, arr_g, arr_a, ptr(gy0),
                                                      ptr(_panel(acc_y0)) if acc_y0 is not None else None, ptr(d.out), ptr(d.ws),
                                                      g.numel(), stream_ptr()))
            dots = (d.lazy(n) if lazy else d.fetch()[:n]) if need_dots else None
        return (gk, dots) if acc_y0 is None else (gk, dots, gy0)

    @staticmethod
    def dot_diff(g, a, b=None, scale=1.0, lazy=False):
        """scale * <g, a - b> (fp64 sum, fixed order): a float, or with lazy a deferred 0-d float32 host tensor (_BwdDots.lazy)"""
        g, a = _panel(g), _panel(a)
        b = _panel(b) if b is not None else None
        d = _BwdDots.get(g.device)
        with _REDUCE_LOCK, torch.cuda.device(g.device):
            check(_lib.load().ndcn_rk_dot_diff_f32(ptr(g), ptr(a), ptr(b), ptr(d.out), ptr(d.ws), g.numel(), stream_ptr()))
            return d.lazy(1, scale)[0] if lazy else d.fetch()[0] * scale

    @staticmethod
    def error_bwd(y0, y1, ks, cs, rtol, atol, g_r, need_y0, need_y1, need_k, need_dots=True, accs=None, acc_y0=None, acc_y1=None, lazy=False):
        """VJP of the error ratio mean(((sum c_j k_j) / tol)^2) for upstream gradient g_r:
        (gy0, gy1, [gk_j], [d ratio / d c_j] (NOT yet multiplied by g_r) as host floats); accs / acc_y0 / acc_y1 as combine_bwd."""
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        arr_k, arr_c, n = _terms(ks, cs)
        gk, arr_g = _grad_ptrs(y0, need_k)
        arr_a = _acc_ptrs(accs, n)
        gy0 = torch.empty_like(y0) if need_y0 else None
        gy1 = torch.empty_like(y0) if need_y1 else None
        d = _BwdDots.get(y0.device)
        with _REDUCE_LOCK, torch.cuda.device(y0.device):
            check(_lib.load().ndcn_rk_error_bwd_f32(ptr(y0), ptr(y1), arr_k, arr_c, n, float(rtol), float(atol), float(g_r),
                                                    1.0 / y0.numel(), ptr(gy0), ptr(gy1), arr_g,
                                                    ptr(_panel(acc_y0)) if acc_y0 is not None else None,
                                                    ptr(_panel(acc_y1)) if acc_y1 is not None else None, arr_a, ptr(d.out), ptr(d.ws),
                                                    y0.numel(), stream_ptr()))
            if lazy:                          # deferred, already multiplied by g_r (the eager form leaves that to the caller)
                return gy0, gy1, gk, (d.lazy(n, float(g_r)) if need_dots else None)
            return gy0, gy1, gk, (d.fetch()[:n] if need_dots else None)

    @staticmethod
    def rms_bwd(a, b, y, rtol, atol, coef, need_a, need_b, need_y):
        """VJP of ||(a - b) / (atol + |y| rtol)|| / sqrt(N); coef = g / (||.|| sqrt(N))."""
        a, y = _panel(a), _panel(y)
        b = _panel(b) if b is not None else None
        ga = torch.empty_like(a) if need_a else None
        gb = torch.empty_like(a) if (need_b and b is not None) else None
        gy = torch.empty_like(a) if need_y else None
        with torch.cuda.device(a.device):
            check(_lib.load().ndcn_rk_rms_bwd_f32(ptr(a), ptr(b), ptr(y), float(rtol), float(atol), float(coef), ptr(ga),
                                                  ptr(gb), ptr(gy), a.numel(), stream_ptr()))
        return ga, gb, gy

    @staticmethod
    def interp_bwd(g, y0, y1, ks, dt, x, need_y0, need_y1, need_k, need_dots=True, accs=None, acc_y0=None, acc_y1=None):
        """VJP of the dopri5 dense output at abscissa x: (gy0, gy1, [gk_j], <g, do/dx>, <g, do/ddt>); accs / acc_y0 / acc_y1 as
        combine_bwd."""
        g, y0, y1 = _panel(g), _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        assert len(ks) == 7
        arr_k = (_P * 7)(*[k.data_ptr() for k in ks])
        gk, arr_g = _grad_ptrs(g, need_k)
        arr_a = _acc_ptrs(accs, 7)
        gy0 = torch.empty_like(g) if need_y0 else None
        gy1 = torch.empty_like(g) if need_y1 else None
        d = _BwdDots.get(g.device)
        with _REDUCE_LOCK, torch.cuda.device(g.device):
            check(_lib.load().ndcn_dopri5_interp_bwd_f32(ptr(g), ptr(y0), ptr(y1), arr_k, float(dt), float(x), ptr(gy0),
                                                         ptr(gy1), arr_g, ptr(_panel(acc_y0)) if acc_y0 is not None else None,
                                                         ptr(_panel(acc_y1)) if acc_y1 is not None else None, arr_a, ptr(d.out),
                                                         ptr(d.ws), g.numel(), stream_ptr()))
            dots = d.fetch() if need_dots else (0.0, 0.0)
        return gy0, gy1, gk, dots[0], dots[1]

    @staticmethod
    def interp_direct_multi(y0, y1, ks, cmid, dt, xpows):
        """len(xpows) <= 8 ticks of one step in one pass: [dense output at each tick] - per tick the arithmetic of interp_direct."""
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        nt = len(xpows)
        assert len(ks) == 7 and 1 <= nt <= 8
        outs = [torch.empty_like(y0) for _ in range(nt)]
        arr_k, arr_c, _ = _terms(ks, cmid)
        xp = (_F * (5 * nt))(*[float(v) for row in xpows for v in row])
        arr_o = (_P * nt)(*[o.data_ptr() for o in outs])
        with torch.cuda.device(y0.device):
            check(_lib.load().ndcn_dopri5_interp_direct_multi_f32(ptr(y0), ptr(y1), arr_k, arr_c, float(dt), xp, arr_o, nt, y0.numel(),
                                                                  stream_ptr()))
        return outs

    @staticmethod
    def interp_bwd_multi(gs, y0, y1, ks, dt, xs, need_y0, need_y1, need_k, accs=None, acc_y0=None, acc_y1=None, lazy=False):
        """VJP of len(gs) <= 7 dense outputs of ONE step: (gy0, gy1, [gk_j], [<g_t, do/dx_t>], sum_t <g_t, do/ddt>)."""
        gs = [_panel(g) for g in gs]
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        nt = len(gs)
        assert len(ks) == 7 and 1 <= nt <= 7 and len(xs) == nt
        arr_k = (_P * 7)(*[k.data_ptr() for k in ks])
        arr_gs = (_P * nt)(*[g.data_ptr() for g in gs])
        arr_x = (_F * nt)(*[float(v) for v in xs])
        gk, arr_g = _grad_ptrs(y0, need_k)
        arr_a = _acc_ptrs(accs, 7)
        gy0 = torch.empty_like(y0) if need_y0 else None
        gy1 = torch.empty_like(y0) if need_y1 else None
        d = _BwdDots.get(y0.device)
        with _REDUCE_LOCK, torch.cuda.device(y0.device):
            check(_lib.load().ndcn_dopri5_interp_bwd_multi_f32(arr_gs, nt, ptr(y0), ptr(y1), arr_k, float(dt), arr_x, ptr(gy0), ptr(gy1), arr_g,
                                                               ptr(_panel(acc_y0)) if acc_y0 is not None else None,
                                                               ptr(_panel(acc_y1)) if acc_y1 is not None else None, arr_a, ptr(d.out),
                                                               ptr(d.ws), y0.numel(), stream_ptr()))
            if lazy:
                dots = d.lazy(8)
                return gy0, gy1, gk, dots[:nt], dots[7]
            dots = d.fetch()
        return gy0, gy1, gk, list(dots[:nt]), dots[7]

    @staticmethod
    def interp_fit(y0, y1, ks, cmid, dt):
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        assert len(ks) == 7
        a, b, c, d = (torch.empty_like(y0) for _ in range(4))
        arr_k, arr_c, _ = _terms(ks, cmid)
        with torch.cuda.device(y0.device):
            check(_lib.load().ndcn_dopri5_interp_fit_f32(ptr(y0), ptr(y1), arr_k, arr_c, float(dt), ptr(a), ptr(b), ptr(c),
                                                         ptr(d), y0.numel(), stream_ptr()))
        return a, b, c, d

    @staticmethod
    def interp_direct(y0, y1, ks, cmid, dt, xpow, out=None):
        """interp_fit + interp_eval in one pass, coefficients not stored (same arithmetic, bit-identical)."""
        y0, y1 = _panel(y0), _panel(y1)
        ks = [_panel(k) for k in ks]
        assert len(ks) == 7
        if out is None:
            out = torch.empty_like(y0)
        arr_k, arr_c, _ = _terms(ks, cmid)
        xp = (_F * 5)(*[float(v) for v in xpow])
        with torch.cuda.device(y0.device):
            check(_lib.load().ndcn_dopri5_interp_direct_f32(ptr(y0), ptr(y1), arr_k, arr_c, float(dt), xp, ptr(out),
                                                            y0.numel(), stream_ptr()))
        return out

    @staticmethod
    def interp_eval(fit, e, xpow, out=None):
        """`fit` = the (a, b, c, d) panels interp_fit returned; e = y at the start of the fitted step."""
        a, b, c, d = fit
        e = _panel(e)
        if out is None:
            out = torch.empty_like(e)
        xp = (_F * 5)(*[float(v) for v in xpow])
        with torch.cuda.device(e.device):
            check(_lib.load().ndcn_interp_eval_f32(ptr(a), ptr(b), ptr(c), ptr(d), ptr(e), xp, ptr(out), e.numel(),
                                                   stream_ptr()))
        return out

    @staticmethod
    def fixed_stage(op, y, k1, k2=None, k3=None, k4=None, dt=0.0, out=None):
        y = _panel(y)
        ks = [None if k is None else _panel(k) for k in (k1, k2, k3, k4)]
        if out is None:
            out = torch.empty_like(y)
        with torch.cuda.device(y.device):
            check(_lib.load().ndcn_fixed_stage_f32(int(op), ptr(out), ptr(y), ptr(ks[0]), ptr(ks[1]), ptr(ks[2]),
                                                   ptr(ks[3]), float(dt), y.numel(), stream_ptr()))
        return out

    # ---------------------------------------------------------------- truth dynamics (N x 1 state)
    @staticmethod
    def row_l1_normalize(X, out=None):
        """F.normalize(X, p=1, dim=1) with infinities zeroed (ode_gcn.py:9-26)."""
        X = _panel(X)
        if out is None:
            out = torch.empty_like(X)
        with torch.cuda.device(X.device):
            check(_lib.load().ndcn_row_l1_normalize_f32(ptr(X), ptr(out), X.shape[0], X.shape[1], stream_ptr()))
        return out

    @staticmethod
    def row_l1_normalize_bwd(g, X):
        """VJP of row_l1_normalize at X for the upstream gradient g."""
        g, X = _panel(g), _panel(X)
        out = torch.empty_like(X)
        with torch.cuda.device(X.device):
            check(_lib.load().ndcn_row_l1_normalize_bwd_f32(ptr(g), ptr(X), ptr(out), X.shape[0], X.shape[1], stream_ptr()))
        return out

    @staticmethod
    def gene_rhs(A, x, b=1.0, f=1.0, h=2.0):
        A = as_csr(A)
        x = _panel(x)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(_lib.load().ndcn_gene_rhs_f32(A.view_ref(), ptr(x), ptr(out), float(b), float(f), float(h), stream_ptr()))
        return out

    @staticmethod
    def mutual_rhs(A, x, b=0.1, k=5., c=1., d=5., e=0.9, h=0.1):
        A = as_csr(A)
        x = _panel(x)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(_lib.load().ndcn_mutual_rhs_f32(A.view_ref(), ptr(x), ptr(out), float(b), float(k), float(c), float(d),
                                                  float(e), float(h), stream_ptr()))
        return out


hip = HipOps()


def new_solve_epoch():
    """Called at the top of every odeint / odeint_adjoint: packed weight images of earlier solves are re-validated by re-packing
    (see _PackedWeights)."""
    _PackedWeights.new_epoch()


def invalidate_packed_weights():
    """Drop the cached packed images of ODEFunc weights (see _PackedWeights): call after writing weights in a way that
    does not move the tensor's version counter (`.data` writes, collectives on `.data`, raw-pointer writes)."""
    _PackedWeights.invalidate()


def device_info():
    out = (ctypes.c_int64 * 6)()
    check(_lib.load().ndcn_device_info(out))
    keys = ('compute_units', 'xcds', 'wave_size', 'clock_khz', 'hbm_mib', 'l2_kib')
    return dict(zip(keys, [int(v) for v in out]))


__all__ = ['HipOps', 'hip', 'device_info', 'CsrOperator', 'as_csr', 'invalidate_packed_weights']
