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
        support = self.fc(input)
        if _needs_grad(support):
            from .autograd_ops import spmm
            return spmm(propagation_adj, support)
        return hip.spmm(propagation_adj, support)


class GCN(nn.Module):
    """models.py:22-47: gc1 -> relu -> [conv_middle -> relu]* -> gc2 with dropout in front of every layer."""

    def __init__(self, input_size, hidden_size, num_classes, dropout=0, num_middle_layers=0):
        super(GCN, self).__init__()

        self.gc1 = GraphConvolution(input_size, hidden_size)
        self.gc2 = GraphConvolution(hidden_size, num_classes)
        self.dropout = dropout

        self.conv_middle = nn.ModuleList([GraphConvolution(hidden_size, hidden_size) for i in range(num_middle_layers)])

    def forward(self, x, propagation_adj):
        x = F.dropout(x, self.dropout, training=self.training)  # drop out for input
        x = self.gc1.forward(x, propagation_adj)
        x = F.relu(x)

        for conv_middle in self.conv_middle:
            x = F.dropout(x, self.dropout, training=self.training)  # drop out for input
            x = conv_middle.forward(x, propagation_adj)
            x = F.relu(x)

        x = F.dropout(x, self.dropout, training=self.training)  # drop out for hidden layers
        x = self.gc2.forward(x, propagation_adj)

        return x
