"""Scene-graph convolution parameter trees (reference model/graph.py:89-250).

Key layout per layer: ``net1.{0,1,3,4}``, ``net2.{0,1,3,4}`` (Linear/BatchNorm1d
pairs from build_mlp), ``linear_projection``, ``linear_projection_pred``.
"""
import torch.nn as nn
from .params import _Holder, Lin, mlp


class GraphTripleConv(_Holder):
    def __init__(self, input_dim_obj, input_dim_pred, output_dim=None, hidden_dim=512,
                 pooling='avg', mlp_normalization='none', residual=True):
        super().__init__()
        if pooling not in ('avg', 'sum'):
            # 'wAvg' exists in the reference (graph.py:105, a learned weighting net) but no shipped config
            # selects it; refuse rather than mis-compute.
            raise NotImplementedError("pooling=%r: 'avg' and 'sum' are on the hot path" % pooling)
        self.pooling = pooling
        output_dim = input_dim_obj if output_dim is None else output_dim
        self.input_dim_obj, self.input_dim_pred = input_dim_obj, input_dim_pred
        self.output_dim, self.hidden_dim, self.residual = output_dim, hidden_dim, residual
        self.net1 = mlp([2 * input_dim_obj + input_dim_pred, hidden_dim,
                         2 * hidden_dim + input_dim_pred], mlp_normalization)
        self.net2 = mlp([hidden_dim, hidden_dim, output_dim], mlp_normalization)
        if residual:
            self.linear_projection = Lin(input_dim_obj, output_dim)
            self.linear_projection_pred = Lin(input_dim_pred, input_dim_pred)


class GraphTripleConvNet(_Holder):
    def __init__(self, input_dim_obj, input_dim_pred, num_layers=2, hidden_dim=512,
                 residual=False, pooling='avg', mlp_normalization='none', output_dim=None):
        super().__init__()
        self.num_layers = num_layers
        self.gconvs = nn.ModuleList()
        for i in range(num_layers):
            # only the last layer changes width (reference graph.py:240-244)
            od = output_dim if (output_dim is not None and i == num_layers - 1) else None
            self.gconvs.append(GraphTripleConv(
                input_dim_obj, input_dim_pred, output_dim=od, hidden_dim=hidden_dim,
                pooling=pooling, mlp_normalization=mlp_normalization, residual=residual))
