"""Seeded synthetic inputs: weights, scene graphs, CLIP-feature stand-ins, noise.

There is no network access for the 3D-FRONT dataset, CLIP or the trained
checkpoints, so benchmarks and parity tests run on
synthetic data with the tensor contract of the reference's ``collate_fn``
(dataset/threedfront_dataset.py:618-743).  Everything here is a pure function of
integer seeds and uses numpy's legacy ``RandomState`` stream, which is stable
across numpy versions and machines -- the golden-vector generator
(tests/golden/make_golden.py) fills the *reference's* modules with the same rule.
"""
import zlib
import numpy as np
import torch


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def seeded_tensor(name, shape, seed=0):
    """Deterministic value for the state-dict entry ``name``.

    Every tensor is non-trivial on purpose: the reference zero-initialises 177
    tensors of the layout UNet (zero_module convs, norm biases) and with stock init
    its output is exactly 0, which would make parity vacuous.
      >=2-D weights : N(0, 1/fan_in)        (keeps activations O(1) through depth)
      1-D 'weight'  : 1 + 0.1 N(0,1)        (GroupNorm / LayerNorm / BatchNorm scale)
      1-D 'bias'    : 0.05 N(0,1)
      running_mean  : 0.1 N(0,1);  running_var : U(0.5, 1.5)
    """
    rs = _rs(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == 'running_var':
        v = rs.uniform(0.5, 1.5, size=shape)
    elif leaf == 'running_mean':
        v = 0.1 * rs.standard_normal(shape)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = rs.standard_normal(shape) / np.sqrt(fan_in)
    elif leaf == 'weight':
        v = 1.0 + 0.1 * rs.standard_normal(shape)
    else:
        v = 0.05 * rs.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


@torch.no_grad()
def seeded_fill_(module_or_sd, seed=0, prefix=''):
    """Overwrite every parameter/buffer of an nn.Module (or a state_dict-like mapping)
    in place with ``seeded_tensor(prefix+name)``.  Returns the object."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, 'state_dict') else module_or_sd
    for k, t in sd.items():
        t.copy_(seeded_tensor(prefix + k, t.shape, seed).to(t.dtype))
    return module_or_sd


def synthetic_graph(num_nodes, seed=0, rel_per_node=3, n_obj_classes=35, n_pred=16):
    """Scene graph with the reference's conventions:
    last node is the ``_scene_`` node (class 0), every other node has an ``in`` edge
    (predicate 0) to it (threedfront_dataset.py:339-350), plus ``rel_per_node * O``
    random pairwise relations with predicate id in [1, n_pred).
    Returns objs int64[O], triples int64[T,3] (s, p, o)."""
    rs = np.random.RandomState(seed)
    O = int(num_nodes)
    objs = rs.randint(1, n_obj_classes, size=O).astype(np.int64)
    objs[-1] = 0
    tri = [(i, 0, O - 1) for i in range(O - 1)]
    if O > 2:
        for _ in range(rel_per_node * O):
            a = int(rs.randint(0, O - 1))
            b = int(rs.randint(0, O - 2))
            if b >= a:
                b += 1
            tri.append((a, int(rs.randint(1, n_pred)), b))
    triples = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    return torch.from_numpy(objs), torch.from_numpy(triples)


def synthetic_features(num_nodes, num_triples, seed=0):
    """Stand-ins for CLIP ViT-B/32 text features: f32[O,512], f32[T,512]."""
    rs = np.random.RandomState(seed + 7919)
    tf = rs.standard_normal((num_nodes, 512)).astype(np.float32)
    rf = rs.standard_normal((num_triples, 512)).astype(np.float32)
    return torch.from_numpy(tf), torch.from_numpy(rf)


def collate_graphs(graphs):
    """Batch several (objs, triples) graphs into one block-diagonal graph with node
    offsets, as the reference's collate_fn does (threedfront_dataset.py:698-701)."""
    objs, tris, off = [], [], 0
    for o, t in graphs:
        objs.append(o)
        t = t.clone()
        t[:, 0] += off
        t[:, 2] += off
        tris.append(t)
        off += o.numel()
    return torch.cat(objs), torch.cat(tris)


def remove_nodes(objs, triples, nodes):
    """The graph without ``nodes`` (editing tests: the encoder side of sample_with_additions sees the scene before the
    objects were added).  Triples touching a removed node are dropped, the remaining indices are compacted.
    Returns objs', triples', kept node indices, kept triple indices (to slice the CLIP features)."""
    O = objs.numel()
    drop = set(int(n) for n in nodes)
    keep_idx = [i for i in range(O) if i not in drop]
    remap = {old: new for new, old in enumerate(keep_idx)}
    tri = triples.tolist()
    keep_tri = [j for j, (s, p, o) in enumerate(tri) if s not in drop and o not in drop]
    new_tri = [(remap[tri[j][0]], tri[j][1], remap[tri[j][2]]) for j in keep_tri]
    return (objs[keep_idx], torch.tensor(new_tri, dtype=torch.int64).reshape(-1, 3),
            torch.tensor(keep_idx, dtype=torch.int64), torch.tensor(keep_tri, dtype=torch.int64))


def layout_noise(num_nodes, box_dim, n_steps, seed=7):
    """Pre-generated noise for loop A: row 0 is x_T, row 1+i is the draw consumed by
    loop iteration i (t = n_steps-1-i), matching the RNG call order of
    p_sample_loop_sg (diffusion_ddpm.py:330-345): 1 + n_steps draws."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((n_steps + 1, num_nodes, box_dim)).astype(np.float32))


def shape_noise(z_shape=(3, 16, 16, 16), seed=7):
    """One latent noise tensor shared by all objects (echo2shape.py:507-510)."""
    rs = np.random.RandomState(seed + 104729)
    return torch.from_numpy(rs.standard_normal((1,) + tuple(z_shape)).astype(np.float32))


VOCAB = {
    # sizes only matter: 35 object names -> Embedding(36, 128); 16 predicates
    'object_idx_to_name': ['obj%02d' % i for i in range(35)],
    'object_idx_to_name_grained': ['obj%02d' % i for i in range(35)],
    'pred_idx_to_name': ['pred%02d' % i for i in range(16)],
}
