"""Oracle-backed test double of the panel-ops interface (ndcn_amd.ops.HipOps).

TEST INFRASTRUCTURE: lets the CPU suite drive the product's host logic
(ndcn_amd/torchdiffeq/_impl/core.py, ndcn_amd/sharding.py) without a GPU.  Never imported by product code.
Every method is the torch-CPU statement of what the corresponding HIP kernel computes, with the
reference's op order (see oracle/ndcn_oracle.py for the reference line numbers).
"""
import numpy as np
import torch

from oracle import ndcn_oracle as orc


def _c(v):
    return torch.tensor(float(v), dtype=torch.float32)


def _wsum(ks, cs):
    acc = 0
    for c, k in zip(cs, ks):
        acc = acc + _c(c) * k
    return acc


def _bad(x):
    return float((~torch.isfinite(x)).sum())


class OracleOps:
    name = 'oracle'
    differentiable = True          # plain torch: the sharded training path differentiates straight through it

    @staticmethod
    def _coo(A):
        if hasattr(A, 'rowptr'):       # ndcn_amd.csr.CsrOperator on the CPU
            return orc.coo_from_csr(A.rowptr.numpy(), A.colidx.numpy(), A.val.numpy(), A.shape)
        return A

    @classmethod
    def spmm(cls, A, X, X_halo=None, alpha=1.0, relu=False, out=None):
        A = cls._coo(A)
        full = X if X_halo is None else torch.cat([X, X_halo], 0)
        y = orc.apply_operator(A, full)
        if alpha != 1.0:
            y = y * alpha
        return torch.relu(y) if relu else y

    @staticmethod
    def linear(S, W, b=None, relu=False):
        y = torch.nn.functional.linear(S, W, b)
        return torch.relu(y) if relu else y

    @classmethod
    def rhs(cls, A, X, W, b, no_graph=False, no_control=False, X_halo=None, out=None):
        x = X
        if not no_graph:
            x = cls.spmm(A, X, X_halo)
        if not no_control:
            x = torch.nn.functional.linear(x, W, b)
        x = torch.relu(x)
        if out is not None:
            out.copy_(x)
            return out
        return x

    _record = [0.0, 0.0]             # the "device" error record of split launches (accum / fetch of HipOps.rhs_rk)

    @classmethod
    def rhs_rk(cls, A, X, W, b, mode, y0, kprev, cs, rtol=0.0, atol=0.0, no_graph=False, no_control=False, X_halo=None,
               out_K=None, out_y=None, y1=None, accum=False, fetch=True, aux_cs=None, out_aux=None, record=None):
        k = cls.rhs(A, X, W, b, no_graph=no_graph, no_control=no_control, X_halo=X_halo)
        ks = list(kprev) + [k]
        if out_K is not None:
            out_K.copy_(k)
        if mode == 'combine':
            y = cls.combine(y0, ks, cs)
            if out_y is not None:
                out_y.copy_(y)
            if aux_cs is not None:
                aux = _wsum(ks, aux_cs)
                if out_aux is not None:
                    out_aux.copy_(aux)
                return k, y, aux
            return k, y
        if mode == 'rk4':
            y = cls.fixed_stage(2 + len(kprev), y0, *ks, dt=cs[0])
            if out_y is not None:
                out_y.copy_(y)
            return k, y
        s, bad = cls.error(y0, X if y1 is None else y1, ks, cs, rtol, atol)
        rec = record if record is not None else cls._record
        rec[:] = [rec[0] + s, rec[1] + bad] if accum else [s, bad]
        return k, (tuple(rec) if fetch else None)

    @staticmethod
    def new_error_record(device):
        return [0.0, 0.0]

    @staticmethod
    def scale(x, w):
        return _c(w) * x

    @staticmethod
    def gather_rows(X, idx):
        return X[idx.long()]

    @staticmethod
    def combine(y0, ks, cs):
        return y0 + _wsum(ks, cs)

    @staticmethod
    def error(y0, y1, ks, cs, rtol, atol):
        err = _wsum(ks, cs)
        tol = atol + rtol * torch.max(torch.abs(y0), torch.abs(y1))
        r = err / tol
        # the float32 sum ATen forms for torch.mean (what the HIP kernel reproduces in ATen's cascade order for panels
        # <= 2^20 elements); fp64 above that, like the product
        v = r * r
        return float(v.sum().double() if 8 <= v.numel() <= (1 << 20) else v.double().sum()), _bad(y1)

    @staticmethod
    def scaled_sumsq(a, b, y, rtol, atol):
        scale = atol + torch.abs(y) * rtol
        q = (a / scale) if b is None else ((a - b) / scale)
        # the float32 norm ATen forms (what the HIP kernel reproduces in ATen's summation order for panels <= 2^20
        # elements): returned as the double whose square root is that norm exactly
        return float(q.norm().double() ** 2), _bad(a)

    @staticmethod
    def interp_fit(y0, y1, ks, cmid, dt):
        dt = _c(dt)
        ymid = y0 + _wsum(ks, cmid)
        f0, f1 = ks[0], ks[-1]
        a = orc._dot([-2 * dt, 2 * dt, -8, -8, 16], [f0, f1, y0, y1, ymid])
        b = orc._dot([5 * dt, -3 * dt, 18, 14, -32], [f0, f1, y0, y1, ymid])
        c = orc._dot([-4 * dt, dt, -11, -5, 16], [f0, f1, y0, y1, ymid])
        d = dt * f0
        return a, b, c, d

    @staticmethod
    def interp_eval(fit, e, xpow, out=None):
        a, b, c, d = fit
        xs = [_c(v) for v in xpow]
        return orc._dot((a, b, c, d, e), xs)

    @staticmethod
    def fixed_stage(op, y, k1, k2=None, k3=None, k4=None, dt=0.0, out=None):
        dt = _c(dt)
        if op == 0:
            return y + dt * k1
        if op == 1:
            return y + k1 * dt / 2
        if op == 2:
            return y + dt * k1 / 3
        if op == 3:
            return y + dt * (k1 / -3 + k2)
        if op == 4:
            return y + dt * (k1 - k2 + k3)
        if op == 5:
            return y + (k1 + 3 * k2 + 3 * k3 + k4) * (dt / 8)
        raise ValueError(op)
