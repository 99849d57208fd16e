"""Device-side versions of the two box helpers the caller applies right after the path
(scripts/eval_3dfront.py:283-284; reference helpers/util.py:542-568) -- SURVEY.md section 8(f) rank 3.
Same names and semantics (``descale_box_params`` writes into its argument and returns it)."""
import ctypes as C
import numpy as np
import torch

from . import hip


def descale_box_params(normed_box_params, file=None, angle=False, stats=None):
    """[-1,1] -> dataset units for sizes (cols 0:3) and translations (cols 3:6), in place on a CUDA tensor."""
    assert file is not None or stats is not None
    if angle:
        raise NotImplementedError('angle=True (7-column boxes) is not used by the sampling path')
    st = np.loadtxt(file) if stats is None else np.asarray(stats)
    x = normed_box_params
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] >= 6 and x.stride(1) == 1):
        raise ValueError('descale_box_params: expects a float32 CUDA tensor [O, >=6]')
    std = torch.tensor(st, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().es_box_postprocess(C.c_void_p(x.data_ptr()), x.stride(0), None, None, C.c_void_p(std.data_ptr()),
                                           x.shape[0], 1.0, hip.current_stream()), 'es_box_postprocess')
    return x


def postprocess_sincos2arctan(sincos):
    """[O,2] (sin, cos) -> [O,1] angle in radians."""
    B, N = sincos.shape
    assert N == 2
    s = sincos.contiguous()
    out = torch.empty(B, 1, dtype=torch.float32, device=s.device)
    hip.check(hip.lib().es_box_postprocess(None, 0, C.c_void_p(s.data_ptr()), C.c_void_p(out.data_ptr()), None, B, 1.0,
                                           hip.current_stream()), 'es_box_postprocess')
    return out
