"""Marching cubes (second half; reference call site model/diff_utils/util_3d.py:214-217).

CPU: the oracle (oracle/mc_oracle.py, table-free loop tracing) on analytic fields, and the product's generated 256-row case
table (echoscene_amd/mc_tables.py) against the oracle's independent polygoniser.
GPU: csrc/es_mc.hip against the oracle on order-independent quantities: the vertex multiset (sorted coordinates), vertex and
face counts, signed volume, area, Euler characteristic, and closedness / consistent orientation -- on an analytic shape, on
a noise field full of ambiguous faces, and on a decoded SDF at the shipped 64^3 resolution with level 0.02.
"""
import numpy as np
import pytest
import torch

from oracle import mc_oracle as mo


def _sphere(n, c, r):
    g = np.mgrid[0:n, 0:n, 0:n].astype(np.float64)
    return np.sqrt(((g - np.asarray(c)[:, None, None, None]) ** 2).sum(0)) - r


def _noise(n, seed):
    rs = np.random.RandomState(seed)
    v = np.full((n + 2,) * 3, 10.0)
    v[1:-1, 1:-1, 1:-1] = rs.standard_normal((n, n, n))
    return v


def test_oracle_sphere_and_noise_invariants():
    v, f = mo.marching_cubes(_sphere(32, (15.3, 16.1, 14.8), 9.7), 0.02)
    inv = mo.mesh_invariants(v, f)
    assert inv['euler'] == 2 and inv['open_or_inconsistent_edges'] == 0
    r = 9.72
    assert abs(-inv['volume'] - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 1e-2      # normals point inside: volume < 0
    assert abs(inv['area'] - 4 * np.pi * r ** 2) / (4 * np.pi * r ** 2) < 1e-2
    v, f = mo.marching_cubes(_noise(20, 0), 0.02)
    inv = mo.mesh_invariants(v, f)
    assert inv['open_or_inconsistent_edges'] == 0, inv                 # ambiguous faces everywhere: still closed + oriented


def test_product_case_table_matches_oracle_polygoniser():
    from echoscene_amd.mc_tables import tri_table
    t = tri_table()
    assert t.shape == (256, 16) and t.dtype == np.int8
    rot = lambda x: tuple(min(tuple(np.roll(x, s)) for s in range(3)))
    for case in range(256):
        mine = t[case][t[case] >= 0].reshape(-1, 3)
        ref = mo._polygonise(case)
        assert sorted(rot(np.asarray(a)) for a in ref) == sorted(rot(a) for a in mine), case
    assert (t[0] < 0).all() and (t[255] < 0).all()


def test_case_table_triangle_counts_are_pinned():
    """The product's case table is generated from a rule (mc_tables.py); its loop structure on ambiguous faces is a CONVENTION that
    differs from PyMCubes' complement-symmetric table.  Pin the per-case triangle counts so it cannot drift unnoticed."""
    import zlib
    from echoscene_amd.mc_tables import tri_table
    t = tri_table()
    cnt = ((t >= 0).sum(1) // 3).astype(np.uint8)
    assert int(cnt.sum()) == 820 and np.bincount(cnt).tolist() == [2, 16, 50, 80, 76, 32]
    assert zlib.crc32(cnt.tobytes()) == 1786192196
    assert (cnt[5], cnt[10], cnt[245], cnt[250]) == (2, 2, 4, 4)          # not complement-symmetric: the classic table has 2, 2, 2, 2
    assert int((cnt != cnt[::-1]).sum()) == 88                             # 44 case / complement pairs differ


def _compare(sdf_np, level):
    from echoscene_amd.postprocess import marching_cubes_batch
    sdf = torch.from_numpy(np.stack(sdf_np).astype(np.float32)).cuda()
    meshes = marching_cubes_batch(sdf, level)
    assert len(meshes) == len(sdf_np)
    for (v, f), vol in zip(meshes, sdf_np):
        assert v.is_cuda and v.dtype == torch.float32 and f.dtype == torch.int64
        rv, rf = mo.marching_cubes(vol.astype(np.float32), level)
        got = mo.mesh_invariants(v.cpu().numpy(), f.cpu().numpy())
        ref = mo.mesh_invariants(rv, rf)
        assert got['V'] == ref['V'] and got['F'] == ref['F'] and got['euler'] == ref['euler'], (got, ref)
        assert got['open_or_inconsistent_edges'] == ref['open_or_inconsistent_edges']
        # vertex multiset: lexicographically sorted coordinates (positions interpolated in fp32 on the device, fp64 in the oracle)
        # vertex multiset: a one-to-one nearest-neighbour matching (positions are interpolated in fp32 on the device, fp64 in
        # the oracle; sorting would be fragile where the interpolated coordinate lands next to a grid point)
        if ref['V']:
            from scipy.spatial import cKDTree
            a = v.cpu().numpy().astype(np.float64)
            dist, _ = cKDTree(rv).query(a)
            dist2, _ = cKDTree(a).query(rv)
            assert dist.max() < 2e-5 and dist2.max() < 2e-5, (dist.max(), dist2.max())      # same set (counts are equal, above)
        assert abs(got['volume'] - ref['volume']) <= 1e-5 * max(1.0, abs(ref['volume']))
        assert abs(got['area'] - ref['area']) <= 1e-5 * max(1.0, ref['area'])
    return meshes


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_marching_cubes_device_vs_oracle_analytic_and_noise():
    _compare([_sphere(32, (15.3, 16.1, 14.8), 9.7), _sphere(32, (10.2, 12.0, 20.5), 6.1)], 0.02)
    _compare([_noise(20, 1), _noise(20, 2), _noise(20, 3)], 0.02)
    _compare([np.full((8, 8, 8), 1.0)], 0.02)                      # empty mesh


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_marching_cubes_on_decoded_sdf_64():
    """The shipped use: the VQ-VAE decoder's [O,1,64,64,64] output at level 0.02 (util_3d.py:194-217), through the
    sdf_to_mesh mirror (verts / n_cell - 0.5)."""
    from echoscene_amd import synth, config as escfg
    from echoscene_amd.model.vqvae import VQVAE
    from echoscene_amd.samplers import VQDecoder
    from echoscene_amd.postprocess import sdf_to_mesh
    c = escfg.vqvae_conf(64).model.params
    vq = VQVAE(dict(c.ddconfig), 8192, c.embed_dim)
    synth.seeded_fill_(vq, prefix='vqvae_full.')
    z = torch.from_numpy((np.random.RandomState(71).standard_normal((2, 3, 16, 16, 16)) * 0.6).astype(np.float32))
    sdf = VQDecoder(vq, torch.device('cuda')).decode_no_quant(z)
    meshes = _compare([sdf[i, 0].cpu().numpy() for i in range(2)], 0.02)
    verts, faces = sdf_to_mesh(sdf, level=0.02)
    assert len(verts) == 2 and torch.allclose(verts[0], meshes[0][0] / 64 - 0.5)
    assert torch.equal(faces[1], meshes[1][1])
