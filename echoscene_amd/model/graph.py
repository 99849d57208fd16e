"""Scene-graph convolution parameter trees (reference model/graph.py:89-250).

Key layout per layer: ``net1.{0,1,3,4}``, ``net2.{0,1,3,4}`` (Linear/BatchNorm1d
pairs from build_mlp), ``linear_projection``, ``linear_projection_pred``, and -- pooling='wAvg' only --
``weightNet.{Net_s,Net_o}.{0,2}``, ``weightNet.down_sample_{obj,pred}``.
"""
import torch.nn as nn
from .params import _Holder, Lin, mlp


class _Head(nn.Sequential):
    """key layout of nn.Sequential(Linear(3 f, 64), ReLU, Linear(64, 1), Sigmoid): parameters under '0' and '2'"""

    def __init__(self, feat_dim):
        super().__init__()
        self.add_module('0', Lin(3 * feat_dim, 64))
        self.add_module('2', Lin(64, 1))

    def forward(self, *a, **k):  # pragma: no cover - parameter container
        raise RuntimeError('parameter container; the hot path runs in the HIP library (echoscene_amd.plan.emit_gcn)')


class WeightNetGCN(_Holder):
    """Parameters of the learned pooling weights of pooling='wAvg' (reference model/graph.py:37-86): one weight per triple
    slot from the down-sampled subject / predicate / object vectors of the triple."""

    def __init__(self, feat_dim_in1=256, feat_dim_in2=256, feat_dim=128, separate_s_o=True):
        super().__init__()
        self.separate = separate_s_o
        if separate_s_o:
            self.Net_s = _Head(feat_dim)
            self.Net_o = _Head(feat_dim)
        else:
            self.Net = _Head(feat_dim)
        self.down_sample_obj = Lin(feat_dim_in1, feat_dim)
        self.down_sample_pred = Lin(feat_dim_in2, feat_dim)


class GraphTripleConv(_Holder):
    def __init__(self, input_dim_obj, input_dim_pred, output_dim=None, hidden_dim=512,
                 pooling='avg', mlp_normalization='none', residual=True):
        super().__init__()
        if pooling not in ('avg', 'sum', 'wAvg'):
            raise ValueError('Invalid pooling "%s"' % pooling)          # the reference asserts (graph.py:105)
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
        if pooling == 'wAvg':
            self.weightNet = WeightNetGCN(hidden_dim, output_dim, 128)


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
