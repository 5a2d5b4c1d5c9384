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


class ODEBlock(nn.Module):
    """reference neural_dynamics.py:42-79."""

    def __init__(self, odefunc, rtol=.01, atol=.001, method='dopri5', adjoint=False, terminal=False):
        super(ODEBlock, self).__init__()
        self.odefunc = odefunc
        self.rtol = rtol
        self.atol = atol
        self.method = method
        self.adjoint = adjoint
        self.terminal = terminal

    def forward(self, vt, x):
        integration_time_vector = vt.type_as(x)
        if self.adjoint:
            out = ode.odeint_adjoint(self.odefunc, x, integration_time_vector,
                                     rtol=self.rtol, atol=self.atol, method=self.method)
        else:
            out = ode.odeint(self.odefunc, x, integration_time_vector,
                             rtol=self.rtol, atol=self.atol, method=self.method)
        return out[-1] if self.terminal else out


class ODEBlock2(nn.Module):
    """reference neural_dynamics.py:82-119 (the time vector is fixed at construction)."""

    def __init__(self, odefunc, vt, rtol=.01, atol=.001, method='dopri5', adjoint=False, terminal=False):
        super(ODEBlock2, self).__init__()
        self.odefunc = odefunc
        self.integration_time_vector = vt
        self.rtol = rtol
        self.atol = atol
        self.method = method
        self.adjoint = adjoint
        self.terminal = terminal

    def forward(self, x):
        integration_time_vector = self.integration_time_vector.type_as(x)
        if self.adjoint:
            out = ode.odeint_adjoint(self.odefunc, x, integration_time_vector,
                                     rtol=self.rtol, atol=self.atol, method=self.method)
        else:
            out = ode.odeint(self.odefunc, x, integration_time_vector,
                             rtol=self.rtol, atol=self.atol, method=self.method)
        return out[-1] if self.terminal else out


class _HipLinear(nn.Linear):
    """nn.Linear whose forward is the MFMA kernel (same parameters / state_dict keys)."""

    def forward(self, x):
        if _needs_grad(x, self):
            from .autograd_ops import linear
            return linear(x, self.weight, self.bias)
        return hip.linear(x, self.weight, self.bias)


class NDCN(nn.Module):
    """encoder -> graph ODE -> decoder   -- reference neural_dynamics.py:122-160."""

    def __init__(self, input_size, hidden_size, A, num_classes, dropout=0.0,
                 no_embed=False, no_graph=False, no_control=False,
                 rtol=.01, atol=.001, method='dopri5'):
        super(NDCN, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.A = A
        self.num_classes = num_classes
        self.dropout = dropout
        self.dropout_layer = nn.Dropout(dropout)
        self.no_embed = no_embed
        self.no_graph = no_graph
        self.no_control = no_control
        self.rtol = rtol
        self.atol = atol
        self.method = method
        self.input_layer = nn.Sequential(_HipLinear(input_size, hidden_size, bias=True), nn.Tanh(),
                                         _HipLinear(hidden_size, hidden_size, bias=True))
        self.neural_dynamic_layer = ODEBlock(
            ODEFunc(hidden_size, A, dropout=dropout, no_graph=no_graph, no_control=no_control),
            rtol=rtol, atol=atol, method=method)
        self.output_layer = _HipLinear(hidden_size, num_classes, bias=True)

    def forward(self, vt, x):
        if not self.no_embed:
            x = self.input_layer(x)
        hvx = self.neural_dynamic_layer(vt, x)
        output = self.output_layer(hvx)
        return output


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
