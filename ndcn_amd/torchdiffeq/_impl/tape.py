"""Backpropagation through a dopri5 solve of a plain ODEFunc as two library calls (csrc/tape.hip).

The reference trains by autograd through odeint (heat_dynamics.py:313-334, dgnn.py:192-222): here `ndcn_tape_dopri5_f32` runs the
solve with the per-operation path's own launches and keeps every attempted step; `ndcn_tape_backward_f32` is its reverse pass - the
VJP kernels of the panel operations plus the hand-written adjoint of the step-size controller's scalar chain (which the reference
differentiates: dt, the initial step, the interpolation abscissa are tensors with history).  One autograd node per solve instead of
~130 (`autograd_path.integrate_dopri5_grad`, which stays the path of tuple states, plain callables, `t` with gradient and the A/B:
NDCN_GRAD_TAPE=0)."""
import ctypes
import os

import numpy as np
import torch
from torch.autograd.function import once_differentiable

from ... import _lib
from ..._lib import check, ptr, stream_ptr

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


def enabled():
    return os.environ.get('NDCN_GRAD_TAPE', '1') != '0' and os.environ.get('NDCN_VJP', 'hip') != 'torch'


class Tape:
    """Owner of one ndcn_tape and of the device memory it asked for (torch's caching allocator, on the solve's stream)."""

    def __init__(self, device):
        self.device = device
        self.blocks = []
        self.error = None
        self.handle = ctypes.c_void_p()
        self.cb = ALLOC_FN(self._alloc)             # (kept: the library calls it until the reverse pass has run)

    def _alloc(self, ctx, nbytes):
        try:
            blk = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        except Exception as e:                      # an exception cannot cross the C frames: the call fails with "no memory", then re-raised
            self.error = e
            return None
        self.blocks.append(blk)
        return blk.data_ptr()

    def close(self):
        if self.handle:
            _lib.load().ndcn_tape_destroy(self.handle)
            self.handle = ctypes.c_void_p()
        self.blocks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _TapeDopri5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, W, b, op, ticks, rtol, atol, opts, step_log):
        csr, csr_t, flags, H = op
        lib = _lib.load()
        y0c = y0.detach().contiguous()
        Wc = W.detach().contiguous() if W is not None else None
        bc = b.detach().contiguous() if b is not None else None
        n_t = len(ticks)
        out = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float32, device=y0c.device)
        tape = Tape(y0c.device)
        tk = (ctypes.c_double * n_t)(*ticks)
        op_arr = (ctypes.c_double * 6)(*opts)
        view = csr.view_ref() if csr is not None else ctypes.byref(_lib.empty_csr(y0c.shape[0]))
        view_t = csr_t.view_ref() if csr_t is not None else None
        with torch.cuda.device(y0c.device):
            rc = lib.ndcn_tape_dopri5_f32(view, view_t, ptr(Wc), ptr(bc), H, flags, ptr(y0c), tk, n_t, float(rtol), float(atol), op_arr,
                                          ptr(out), ctypes.cast(tape.cb, ctypes.c_void_p), None, ctypes.byref(tape.handle), stream_ptr())
        if step_log is not None and tape.handle:
            n = int(lib.ndcn_tape_steplog(tape.handle, None, 0))
            rows = (ctypes.c_double * (5 * max(n, 1)))()
            lib.ndcn_tape_steplog(tape.handle, rows, n)
            step_log.extend(tuple(rows[5 * i + j] for j in range(5)) for i in range(n))
            step_log.append(('nfe', int(lib.ndcn_tape_nfe(tape.handle))))
        if rc < 0:
            text = lib.ndcn_last_error().decode('utf-8', 'replace')
            err = tape.error
            tape.close()
            if err is not None:
                raise err
            if rc in (_lib.EMAXSTEPS, _lib.EUNDERFLOW, _lib.ENONFINITE):
                raise AssertionError(text)          # the reference's assertions (dopri5.py:89,100-102)
            raise _lib.NdcnHipError(rc, text)
        ctx.tape, ctx.keep = tape, (y0c, Wc, bc, csr, csr_t)
        ctx.has = (W is not None, b is not None)
        # (the library reads y0 / W / b again in the reverse pass: saved through autograd, so that an in-place change between forward
        # and backward raises instead of giving a gradient at other parameters - round-4 advisor on the fixed-grid Functions)
        # The forward record's device blocks ride along as saved tensors: autograd releases them with the graph - after the first
        # backward unless retain_graph - and a reverse pass over a released graph raises autograd's own "second time" error; with
        # retain_graph the record stays and the pass runs again (round-5 advisor: the reference's graph is re-runnable)
        blocks, tape.blocks = tape.blocks, []
        ctx.save_for_backward(y0, W, b, *blocks)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        tape = ctx.tape
        record = ctx.saved_tensors                  # version check of y0 / W / b; raises once the graph (and with it the record) is released
        if tape is None or not tape.handle:
            raise RuntimeError('the dopri5 tape of this solve has been destroyed')
        y0c, Wc, bc, csr, csr_t = ctx.keep
        g = g.contiguous()
        gy = torch.empty_like(y0c)
        gW = torch.empty_like(Wc) if Wc is not None else None
        gb = torch.empty_like(bc) if bc is not None else None
        lib = _lib.load()
        try:
            with torch.cuda.device(g.device):
                rc = lib.ndcn_tape_backward_f32(tape.handle, ptr(g), ptr(gy), ptr(gW), ptr(gb), stream_ptr())
            if rc < 0:
                if tape.error is not None:
                    raise tape.error
                check(rc)
        finally:
            tape.blocks = []                        # this pass's scratch (kernels still queued read it: the caching allocator reuses blocks stream-ordered)
            del record
        needs = ctx.needs_input_grad
        return (gy if needs[0] else None, gW if (needs[1] and ctx.has[0]) else None, gb if (needs[2] and ctx.has[1]) else None,
                None, None, None, None, None, None)


def applicable(odefunc, y0, t_user):
    if not enabled() or (torch.is_tensor(t_user) and t_user.requires_grad):
        return False
    if y0.dim() != 2 or y0.dtype != torch.float32:
        return False
    if not odefunc.no_graph:
        from ...csr import as_csr
        A = as_csr(odefunc.A)
        if A.shape[0] != A.shape[1] or A.shape[0] != y0.shape[0] or A.device != y0.device:
            return False
    return True


def solve(odefunc, y0, t, rtol, atol, options, step_log):
    """-> trajectory (T, N, H) with one autograd node, or None where the tape does not apply"""
    from . import core
    from ...csr import as_csr
    opt = core.dopri5_options(options, 1)
    rt, at = core.per_state_tolerance(rtol, 1)[0], core.per_state_tolerance(atol, 1)[0]
    flags = _lib.F_RELU | (_lib.F_NO_GRAPH if odefunc.no_graph else 0) | (_lib.F_NO_CONTROL if odefunc.no_control else 0)
    csr = csr_t = None
    if not odefunc.no_graph:
        csr = as_csr(odefunc.A)
        csr.ensure_plans(odefunc.hidden_size)
        csr_t = csr.transpose()
        csr_t.ensure_plans(odefunc.hidden_size)
    W = b = None
    if not odefunc.no_control:
        W, b = odefunc.wt.weight, odefunc.wt.bias
    ticks = core.host_grid(t).to(torch.float64).tolist()
    from .autograd_path import _keep_s_enabled
    keep_s = (not odefunc.no_graph) and (not odefunc.no_control) and _keep_s_enabled(y0)      # (the library keeps S where a kernel writes it)
    opts = (0.0 if opt['first_step'] is None else 1.0, opt['safety'], opt['ifactor'], opt['dfactor'], float(min(opt['max_num_steps'], 2 ** 53)),
            1.0 if keep_s else 0.0)
    return _TapeDopri5.apply(y0, W, b, (csr, csr_t, flags, odefunc.hidden_size), ticks, rt, at, opts, step_log)


# ---- fixed grids ---------------------------------------------------------------------------------------------------------------

class _NativeFixedGrid(torch.autograd.Function):
    """Euler / midpoint / RK4 over ODEFunc with the loops of `_impl/odeint.py::_FixedGridSolve` inside the library
    (ndcn_fixed_grid_train_f32 / ndcn_fixed_grid_backward_f32): the same launches, two calls instead of ~10 per step."""

    @staticmethod
    def forward(ctx, y0, W, b, csr, flags, method, dts):
        lib = _lib.load()
        y0c = y0.detach().contiguous()
        Wc = W.detach().contiguous() if W is not None else None
        bc = b.detach().contiguous() if b is not None else None
        n_ticks = len(dts)
        out = torch.empty((n_ticks + 1,) + tuple(y0c.shape), dtype=torch.float32, device=y0c.device)
        scratch = Tape(y0c.device)
        arr = (ctypes.c_float * n_ticks)(*dts)
        view = csr.view_ref() if csr is not None else ctypes.byref(_lib.empty_csr(y0c.shape[0]))
        H = y0c.shape[1]
        with torch.cuda.device(y0c.device):
            rc = lib.ndcn_fixed_grid_train_f32(view, ptr(Wc), ptr(bc), H, flags, _lib.METHODS[method], ptr(y0c), arr, n_ticks, ptr(out),
                                               ctypes.cast(scratch.cb, ctypes.c_void_p), None, stream_ptr())
        err = scratch.error
        scratch.close()
        if rc < 0:
            if err is not None:
                raise err
            check(rc)
        ctx.keep = (Wc, bc, csr, flags, method, arr, n_ticks)
        ctx.has = (W is not None, b is not None)
        # `out` is this node's own output: saved through autograd (no out -> grad_fn -> ctx -> out cycle that only the cyclic collector
        # would break - a (T, N, H) trajectory - and an in-place edit of the returned trajectory before backward raises)
        ctx.save_for_backward(out, W, b)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        out = ctx.saved_tensors[0]
        Wc, bc, csr, flags, method, arr, n_ticks = ctx.keep
        lib = _lib.load()
        g = g.contiguous()
        no_control = bool(flags & _lib.F_NO_CONTROL)
        gy = torch.empty_like(out[0])
        gW = torch.empty_like(Wc) if (Wc is not None and not no_control) else None
        gb = torch.empty_like(bc) if (bc is not None and not no_control) else None
        scratch = Tape(g.device)
        view = csr.view_ref() if csr is not None else ctypes.byref(_lib.empty_csr(out.shape[1]))
        view_t = csr.transpose().view_ref() if csr is not None else None
        with torch.cuda.device(g.device):
            rc = lib.ndcn_fixed_grid_backward_f32(view, view_t, ptr(Wc), ptr(bc), out.shape[2], flags, _lib.METHODS[method], ptr(out), ptr(g), arr,
                                                  n_ticks, ptr(gy), ptr(gW), ptr(gb), ctypes.cast(scratch.cb, ctypes.c_void_p), None, stream_ptr())
        err = scratch.error
        scratch.close()
        if rc < 0:
            if err is not None:
                raise err
            check(rc)
        needs = ctx.needs_input_grad
        return (gy if needs[0] else None, gW if needs[1] else None, gb if needs[2] else None, None, None, None, None)


def fixed_grid(y0, W, b, csr, flags, method, dts):
    """-> trajectory (T, N, H), or None when the native loops are switched off (NDCN_FIXED_GRID_NATIVE=0)"""
    if os.environ.get('NDCN_FIXED_GRID_NATIVE', '1') == '0' or os.environ.get('NDCN_VJP', 'hip') == 'torch':
        return None
    if csr is not None:
        csr.ensure_plans(y0.shape[1])
        csr.transpose().ensure_plans(y0.shape[1])
    return _NativeFixedGrid.apply(y0, W, b, csr, flags, method, dts)
