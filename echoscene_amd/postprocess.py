"""Device-side versions of the two box helpers the caller applies right after the path
(scripts/eval_3dfront.py:283-284; reference helpers/util.py:542-568) -- SURVEY.md section 8(f) rank 3.
Same names and semantics (``descale_box_params`` writes into its argument and returns it)."""
import ctypes as C
import numpy as np
import torch

from . import hip


def descale_box_params(normed_box_params, file=None, angle=False, stats=None):
    """[-1,1] -> dataset units for sizes (cols 0:3) and translations (cols 3:6), in place on a CUDA tensor; ``angle=True``: also
    column 6, a normalised angle, to [stats[12], stats[13]] (helpers/util.py:553-555)."""
    assert file is not None or stats is not None
    ncol = 7 if angle else 6
    st = np.loadtxt(file) if stats is None else np.asarray(stats)
    if st.size < (14 if angle else 12):
        raise ValueError('descale_box_params: %d statistics given, %d needed%s' % (st.size, 14 if angle else 12, ' (angle=True reads entries 12, 13)' if angle else ''))
    x = normed_box_params
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] >= ncol and x.stride(1) == 1):
        raise ValueError('descale_box_params: expects a float32 CUDA tensor [O, >=%d]' % ncol)
    std = torch.tensor(st, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().es_box_descale(C.c_void_p(x.data_ptr()), x.stride(0), ncol, C.c_void_p(std.data_ptr()), x.shape[0],
                                       hip.current_stream()), 'es_box_descale')
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


_MC_TABLE = {}


def _mc_table(device):
    from .mc_tables import tri_table
    key = str(device)
    if key not in _MC_TABLE:
        _MC_TABLE[key] = torch.from_numpy(tri_table().copy()).to(device)
    return _MC_TABLE[key]


def marching_cubes_batch(sdf, level=0.02):
    """SDF grids -> indexed triangle meshes on the device (csrc/es_mc.hip).

    ``sdf``: float32 CUDA tensor [O, n, n, n] (or [O, 1, n, n, n], the layout ``rel2shape`` returns).  Returns a list of
    (verts f32 [V,3] in grid-index units, faces int64 [T,3]) per object, CUDA tensors -- the values
    ``mcubes.marching_cubes(sdf[i, 0].cpu().numpy(), level)`` produces in the reference (util_3d.py:214-217) for the VERTEX SET
    (one vertex per sign-changing grid edge, same interpolation), up to their order.  The FACES follow this package's own case
    table (mc_tables.py): the diagonals inside a cell AND, on ambiguous-face cases, the loop structure / triangle count differ
    from PyMCubes' classic table (44 case / complement pairs) -- face counts and topology can differ there; meshes are
    watertight either way."""
    if sdf.dim() == 5:
        assert sdf.shape[1] == 1
        sdf = sdf[:, 0]
    if not (sdf.is_cuda and sdf.dim() == 4 and sdf.shape[1] == sdf.shape[2] == sdf.shape[3]):
        raise ValueError('marching_cubes_batch: expects a CUDA tensor [O, n, n, n]')
    sdf = sdf.contiguous().float()
    O, n = sdf.shape[0], sdf.shape[1]
    dev = sdf.device
    L = hip.lib()
    tab = _mc_table(dev)
    ws = torch.empty(L.es_marching_cubes_workspace(O, n), dtype=torch.uint8, device=dev)
    counts = torch.empty(2 * O, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    hip.check(L.es_marching_cubes_count(p(sdf), O, n, float(level), p(tab), p(ws), p(counts), hip.current_stream()),
              'es_marching_cubes_count')
    c = counts.cpu().view(O, 2).long()                     # the one host round trip: output sizes are data dependent
    vofs = torch.cat([torch.zeros(1, dtype=torch.long), c[:, 0].cumsum(0)])
    tofs = torch.cat([torch.zeros(1, dtype=torch.long), c[:, 1].cumsum(0)])
    verts = torch.empty(max(int(vofs[-1]), 1), 3, dtype=torch.float32, device=dev)
    faces = torch.empty(max(int(tofs[-1]), 1), 3, dtype=torch.int32, device=dev)
    vo, to = vofs[:-1].contiguous().to(dev), tofs[:-1].contiguous().to(dev)
    hip.check(L.es_marching_cubes_emit(p(sdf), O, n, float(level), p(tab), p(ws), p(vo), p(to), p(verts), p(faces),
                                       hip.current_stream()), 'es_marching_cubes_emit')
    return [(verts[int(vofs[o]):int(vofs[o + 1])], faces[int(tofs[o]):int(tofs[o + 1])].long()) for o in range(O)]


def marching_cubes(sdf_i, level):
    """Same call shape as ``mcubes.marching_cubes(volume, isovalue)`` for one [n,n,n] CUDA grid -> (verts, faces)."""
    return marching_cubes_batch(sdf_i[None], level)[0]


def sdf_to_mesh(sdf, level=0.02, render_all=False):
    """util_3d.sdf_to_mesh (model/diff_utils/util_3d.py:194-236) up to the pytorch3d container: per-object vertex lists
    normalised like the reference (``verts / n_cell - 0.5``) and int64 faces, at most 16 objects unless ``render_all``."""
    bs, n_cell = sdf.shape[0], sdf.shape[-1]
    k = bs if render_all else min(bs, 16)
    meshes = marching_cubes_batch(sdf[:k], level)
    return [v / n_cell - 0.5 for v, _ in meshes], [f for _, f in meshes]
