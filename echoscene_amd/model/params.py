"""Parameter containers that reproduce the reference's ``state_dict`` key layout.

The reference keeps its weights in ``torch.nn`` module trees and checkpoints
them with ``state_dict()`` (``model/SGDiff.py:49-84``, ``model/EchoScene.py:534-543``).
To accept those checkpoints unchanged, the host side of this build keeps module
trees with *identical attribute names and tensor shapes* -- but they are pure
parameter holders: none of them has a ``forward``.  All arithmetic on the hot
path is done by the HIP library (``echoscene_amd/csrc``) driven from
``echoscene_amd/plan.py``; calling one of these holders raises, so there is no
way to fall back to a PyTorch implementation by accident.

Tensors are allocated uninitialised (``torch.empty``); callers either load a
checkpoint or fill them with ``echoscene_amd.synth.seeded_fill_``.
"""
import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard, never on the product path
        raise RuntimeError(
            "%s is a parameter container; the hot path runs in the HIP library "
            "(echoscene_amd.plan), not in PyTorch" % type(self).__name__)


def _p(*shape):
    return nn.Parameter(torch.empty(*shape), requires_grad=False)


class Lin(_Holder):
    """nn.Linear-shaped holder: weight [out,in], optional bias [out]."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = _p(cout, cin)
        if bias:
            self.bias = _p(cout)
        else:
            self.register_parameter('bias', None)


class Conv(_Holder):
    """nn.ConvNd-shaped holder: weight [out,in,k,...] (dims = 1 or 3)."""

    def __init__(self, dims, cin, cout, k, bias=True):
        super().__init__()
        self.weight = _p(cout, cin, *([k] * dims))
        if bias:
            self.bias = _p(cout)
        else:
            self.register_parameter('bias', None)


class Affine(_Holder):
    """GroupNorm / LayerNorm holder: weight [C], bias [C]."""

    def __init__(self, ch):
        super().__init__()
        self.weight = _p(ch)
        self.bias = _p(ch)


class BatchNormStats(_Holder):
    """nn.BatchNorm1d holder (eval-mode statistics are what sampling uses,
    SURVEY.md section 7 hard part (4))."""

    def __init__(self, ch):
        super().__init__()
        self.weight = _p(ch)
        self.bias = _p(ch)
        self.register_buffer('running_mean', torch.empty(ch))
        self.register_buffer('running_var', torch.empty(ch))
        self.register_buffer('num_batches_tracked', torch.zeros((), dtype=torch.long))


class Emb(_Holder):
    def __init__(self, n, d):
        super().__init__()
        self.weight = _p(n, d)


class Slot(_Holder):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the
    reference (SiLU / ReLU / Dropout / MaxPool / Flatten / Identity)."""


def seq(*mods):
    return nn.Sequential(*mods)


def mlp(dims, batch_norm, final_nonlinearity=True):
    """Index layout of ``build_mlp`` (reference model/layers.py:21-38):
    Linear, [BatchNorm1d], ReLU per layer; the last layer keeps norm+ReLU only when
    ``final_nonlinearity``."""
    mods = []
    for i in range(len(dims) - 1):
        mods.append(Lin(dims[i], dims[i + 1]))
        last = i == len(dims) - 2
        if not last or final_nonlinearity:
            if batch_norm == 'batch':
                mods.append(BatchNormStats(dims[i + 1]))
            mods.append(Slot())
    return seq(*mods)
