"""Drop-in for the residual-GCN pieces of the reference's ode_gcn.py: `row_normalization`,
`RowNorm`, `ResBlock` - same constructor signatures, attribute names (`.A`, `.linear`, `.time_step`,
`.dropout_layer`) and state_dict keys, computed by libndcn_hip.so:

    ResBlock.forward (ode_gcn.py:48-60)      x + relu([rownorm]([W](A [rownorm] x))) * time_step
      A x           ndcn_spmm_f32       (ReLU fused into the SpMM when nothing sits between them)
      W f + b       ndcn_linear_f32     (time_varying)
      rownorm       ndcn_row_l1_normalize_f32
      x + f * ts    ndcn_fixed_stage_f32 op 0 (y + dt k: the product is rounded before the sum, as the reference's)

The commented-out ODEFunc / ODEBlock variants of that file are dead code in the reference; the live ones are in
neural_dynamics.py.  With gradients enabled (training) the SpMM / Linear run through the autograd Functions of
autograd_ops (HIP forward, VJP = SpMM with A^T / GEMM) and the pointwise parts through torch on the device.
"""
import torch
import torch.nn as nn

from .neural_dynamics import _needs_grad
from .ops import hip


class _RowNorm(torch.autograd.Function):
    """ndcn_row_l1_normalize_f32 forward, ndcn_row_l1_normalize_bwd_f32 backward."""

    @staticmethod
    def forward(ctx, X):
        ctx.save_for_backward(X)
        return hip.row_l1_normalize(X)

    @staticmethod
    def backward(ctx, g):
        (X,) = ctx.saved_tensors
        return hip.row_l1_normalize_bwd(g.contiguous(), X)


def row_normalization(X):
    """Row-normalize: each row / max(L1 norm, 1e-12), infinities zeroed  (ode_gcn.py:9-16).  Device tensors only (host
    tensors are refused like everywhere in this package: there is no CPU path); with gradients enabled the backward is
    the library's VJP kernel."""
    X = X.float()
    if _needs_grad(X):
        from ._lib import require_device
        return _RowNorm.apply(require_device(X, 'row_normalization input').contiguous())
    return hip.row_l1_normalize(X)


class RowNorm(nn.Module):
    """ode_gcn.py:19-26."""

    def __init__(self):
        super(RowNorm, self).__init__()

    def forward(self, X):
        return row_normalization(X)


class ResBlock(nn.Module):
    """ode_gcn.py:29-60."""

    def __init__(self, hidden_size, A, dropout=0, normalize=False, time_varying=False, Euler=False):
        super().__init__()
        self.hidden_size, self.A = hidden_size, A
        self.dropout, self.dropout_layer = dropout, nn.Dropout(dropout)
        self.normalize, self.time_varying, self.Euler = normalize, time_varying, Euler
        if time_varying:                     # per-block weight: state_dict keys linear.weight / linear.bias
            self.linear = nn.Linear(hidden_size, hidden_size, bias=True)
        # learnable step of the residual update, drawn from U(0, 1) like the reference (ode_gcn.py:43-45); plain 1 otherwise
        self.time_step = nn.Parameter(torch.empty(1).uniform_(0, 1)) if Euler else 1

    def forward(self, x):
        shortcut = x
        stochastic = self.dropout > 0 and self.training
        if _needs_grad(x, self) or stochastic:
            from .autograd_ops import spmm, linear
            if self.normalize:
                x = row_normalization(x)
            f = spmm(self.A, x)
            if self.time_varying:
                f = linear(f, self.linear.weight, self.linear.bias)
            f = self.dropout_layer(f)
            if self.normalize:
                f = row_normalization(f)
            return shortcut + torch.relu(f) * self.time_step
        if self.normalize:
            x = hip.row_l1_normalize(x)
        plain = not (self.time_varying or self.normalize)
        f = hip.spmm(self.A, x, relu=plain)
        if self.time_varying:
            f = hip.linear(f, self.linear.weight, self.linear.bias, relu=not self.normalize)
        if self.normalize:
            f = torch.relu_(hip.row_l1_normalize(f, out=f))
        ts = float(self.time_step) if not torch.is_tensor(self.time_step) else float(self.time_step.detach())
        return hip.fixed_stage(0, shortcut, f, dt=ts)
