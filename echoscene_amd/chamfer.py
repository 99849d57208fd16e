"""Drop-in for the reference's ``extension/old_chamfer/dist_chamfer.py``: ``chamferDist()(a, b) -> (dist1, dist2)``
with autograd, on the HIP kernels of csrc/es_chamfer.hip (used by scripts/consistency_check.py:78-88)."""
import ctypes as C
import torch
from torch.autograd import Function

from . import hip


class chamferFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        if not (xyz1.is_cuda and xyz2.is_cuda):
            raise RuntimeError('chamferDist: GPU tensors only (as the reference)')
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist1 = torch.zeros(b, n, device=xyz1.device)
        dist2 = torch.zeros(b, m, device=xyz1.device)
        idx1 = torch.zeros(b, n, dtype=torch.int32, device=xyz1.device)
        idx2 = torch.zeros(b, m, dtype=torch.int32, device=xyz1.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        hip.check(hip.lib().es_chamfer_forward(p(xyz1), p(xyz2), b, n, m, p(dist1), p(idx1), p(dist2), p(idx2),
                                               hip.current_stream()), 'es_chamfer_forward')
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1, graddist2 = graddist1.contiguous(), graddist2.contiguous()
        g1, g2 = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        p = lambda t: C.c_void_p(t.data_ptr())
        hip.check(hip.lib().es_chamfer_backward(p(xyz1), p(xyz2), b, n, m, p(graddist1), p(graddist2), p(idx1), p(idx2),
                                                p(g1), p(g2), hip.current_stream()), 'es_chamfer_backward')
        return g1, g2


class chamferDist(torch.nn.Module):
    def forward(self, input1, input2):
        return chamferFunction.apply(input1, input2)
