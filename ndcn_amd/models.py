"""Drop-in for the sparse GraphConvolution of the reference's models.py:8-20 (Linear first, then SpMM) and the
GCN stack built from it (models.py:22-47).  The DeepGCN* baseline classes of that file are out
of scope."""
import torch.nn as nn
import torch.nn.functional as F

from .neural_dynamics import _HipLinear, _needs_grad
from .ops import hip


class GraphConvolution(nn.Module):

    def __init__(self, input_size, output_size, bias=True):
        super(GraphConvolution, self).__init__()
        self.fc = _HipLinear(input_size, output_size, bias=bias)

    def forward(self, input, propagation_adj):
        if not _needs_grad(input, self.fc):
            return hip.gcn(propagation_adj, input, self.fc.weight, self.fc.bias)      # ndcn_gcn_f32: Linear -> SpMM in one call
        from .autograd_ops import spmm
        return spmm(propagation_adj, self.fc(input))


class GCN(nn.Module):
    """The GCN stack of models.py:22-47: dropout -> gc1 -> relu, `num_middle_layers` times dropout -> conv_middle[i] ->
    relu, then dropout -> gc2 (logits, no activation).  Submodule names gc1 / gc2 / conv_middle are the reference's
    state_dict keys."""

    def __init__(self, input_size, hidden_size, num_classes, dropout=0, num_middle_layers=0):
        super().__init__()
        self.dropout = dropout
        self.gc1 = GraphConvolution(input_size, hidden_size)
        self.gc2 = GraphConvolution(hidden_size, num_classes)
        self.conv_middle = nn.ModuleList(GraphConvolution(hidden_size, hidden_size) for _ in range(num_middle_layers))

    def forward(self, x, propagation_adj):
        hidden_layers = [self.gc1] + list(self.conv_middle)
        for layer in hidden_layers:
            x = F.relu(layer(F.dropout(x, self.dropout, training=self.training), propagation_adj))
        return self.gc2(F.dropout(x, self.dropout, training=self.training), propagation_adj)
