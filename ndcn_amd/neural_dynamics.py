"""Drop-in for the hot-path classes of the reference's neural_dynamics.py: ODEFunc, ODEBlock, ODEBlock2,
NDCN, GraphConvolution - same constructor signatures, attribute names and state_dict keys
(SURVEY.md 8b), computed by the HIP kernels of libndcn_hip.so.

Out of scope: TemporalGCN (discrete RNN baselines, neural_dynamics.py:179-238).
"""
import torch
import torch.nn as nn

from . import torchdiffeq as ode
from .ops import hip


def _needs_grad(*tensors_and_modules):
    if not torch.is_grad_enabled():
        return False
    for obj in tensors_and_modules:
        if isinstance(obj, nn.Module):
            if any(p.requires_grad for p in obj.parameters()):
                return True
        elif torch.is_tensor(obj) and obj.requires_grad:
            return True
    return False


class ODEFunc(nn.Module):
    """dX/dt = relu(dropout(W (A X) + b))   -- reference neural_dynamics.py:8-39."""

    ndcn_autonomous = True        # forward ignores t (neural_dynamics.py:20-26)

    def __init__(self, hidden_size, A, dropout=0.0, no_graph=False, no_control=False):
        super(ODEFunc, self).__init__()
        self.hidden_size = hidden_size
        self.dropout = dropout
        self.dropout_layer = nn.Dropout(dropout)
        self.A = A  # N_node * N_node: dense tensor, torch sparse tensor or ndcn_amd.CsrOperator
        self.wt = nn.Linear(hidden_size, hidden_size)
        self.no_graph = no_graph
        self.no_control = no_control

    def forward(self, t, x):
        if self.dropout > 0 and self.training:
            # stochastic RHS: un-fused sequence, dropout between Linear and relu as neural_dynamics.py:34
            from .autograd_ops import spmm, linear
            if not self.no_graph:
                x = spmm(self.A, x)
            if not self.no_control:
                x = linear(x, self.wt.weight, self.wt.bias)
            x = self.dropout_layer(x)
            return torch.relu(x)
        if _needs_grad(x, self.wt):
            from .autograd_ops import rhs
            return rhs(self.A, x, self.wt.weight, self.wt.bias, self.no_graph, self.no_control)
        return hip.rhs(self.A, x, self.wt.weight, self.wt.bias, no_graph=self.no_graph, no_control=self.no_control)


class _OdeBlock(nn.Module):
    """What the reference's two block classes share (neural_dynamics.py:42-119): a right-hand side module held as
    `.odefunc` (the state_dict prefix `odefunc.`), the solver settings as plain attributes, and one solve per forward -
    `odeint_adjoint` when `adjoint` is set, the whole trajectory or its last tick."""

    def _configure(self, odefunc, rtol, atol, method, adjoint, terminal):
        self.odefunc = odefunc
        self.rtol, self.atol, self.method = rtol, atol, method
        self.adjoint, self.terminal = adjoint, terminal

    def _solve(self, x, vt):
        solve = ode.odeint_adjoint if self.adjoint else ode.odeint
        out = solve(self.odefunc, x, vt.type_as(x), rtol=self.rtol, atol=self.atol, method=self.method)
        return out[-1] if self.terminal else out


class ODEBlock(_OdeBlock):
    """forward(vt, x): the time vector arrives with every call (neural_dynamics.py:42-79; NDCN's block)."""

    def __init__(self, odefunc, rtol=.01, atol=.001, method='dopri5', adjoint=False, terminal=False):
        super().__init__()
        self._configure(odefunc, rtol, atol, method, adjoint, terminal)

    def forward(self, vt, x):
        return self._solve(x, vt)


class ODEBlock2(_OdeBlock):
    """forward(x): the time vector is fixed at construction as `.integration_time_vector` - a plain attribute, not a
    buffer, so `.to(device)` leaves it where the caller put it (neural_dynamics.py:82-119; dgnn.py's Sequential)."""

    def __init__(self, odefunc, vt, rtol=.01, atol=.001, method='dopri5', adjoint=False, terminal=False):
        super().__init__()
        self._configure(odefunc, rtol, atol, method, adjoint, terminal)
        self.integration_time_vector = vt

    def forward(self, x):
        return self._solve(x, self.integration_time_vector)


class _HipLinear(nn.Linear):
    """nn.Linear whose forward is the MFMA kernel (same parameters / state_dict keys)."""

    def forward(self, x):
        if _needs_grad(x, self):
            from .autograd_ops import linear
            return linear(x, self.weight, self.bias)
        return hip.linear(x, self.weight, self.bias)


class NDCN(nn.Module):
    """encoder -> graph ODE -> decoder   -- reference neural_dynamics.py:122-160.  Submodule names (and with them the
    state_dict keys input_layer.{0,2}.*, neural_dynamic_layer.odefunc.wt.*, output_layer.*) and the constructor's
    arguments, kept as attributes, are the contract; the three Linears run on the MFMA kernel."""

    def __init__(self, input_size, hidden_size, A, num_classes, dropout=0.0,
                 no_embed=False, no_graph=False, no_control=False,
                 rtol=.01, atol=.001, method='dopri5'):
        super().__init__()
        for name, value in (('input_size', input_size), ('hidden_size', hidden_size), ('A', A), ('num_classes', num_classes),
                            ('dropout', dropout), ('no_embed', no_embed), ('no_graph', no_graph), ('no_control', no_control),
                            ('rtol', rtol), ('atol', atol), ('method', method)):
            setattr(self, name, value)
        self.dropout_layer = nn.Dropout(dropout)
        # construction order = parameter order = the order nn.Linear draws its initial weights in (seeded parity)
        self.input_layer = nn.Sequential(_HipLinear(input_size, hidden_size), nn.Tanh(), _HipLinear(hidden_size, hidden_size))
        func = ODEFunc(hidden_size, A, dropout=dropout, no_graph=no_graph, no_control=no_control)
        self.neural_dynamic_layer = ODEBlock(func, rtol=rtol, atol=atol, method=method)
        self.output_layer = _HipLinear(hidden_size, num_classes)

    def forward(self, vt, x):
        h = x if self.no_embed else self.input_layer(x)
        return self.output_layer(self.neural_dynamic_layer(vt, h))


class GraphConvolution(nn.Module):
    """A (x W^T + b) flattened to 1 x (N*out)   -- reference neural_dynamics.py:163-176 (dense-A variant
    that dgnn.py's star-import exposes)."""

    def __init__(self, input_size, output_size, bias=True):
        super(GraphConvolution, self).__init__()
        self.fc = _HipLinear(input_size, output_size, bias=bias)

    def forward(self, input, propagation_adj):
        support = self.fc(input)
        if _needs_grad(support):
            from .autograd_ops import spmm
            output = spmm(propagation_adj, support)
        else:
            output = hip.spmm(propagation_adj, support)
        return output.view(1, -1)
