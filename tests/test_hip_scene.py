"""GPU parity of the drop-in boundary: the SGDiff / Sg2ScDiffModel / Sg2BoxDiffModel mirror on the
HIP path vs golden vectors produced by the reference's own ``SGDiff`` API, plus the VQ-VAE decode
epilogue."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from echoscene_amd import synth, config as escfg

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


@pytest.mark.parametrize('tag', ['tiny', 'full'])
def test_vqvae_decode_vs_reference_golden(tag):
    from echoscene_amd.model.vqvae import VQVAE
    from echoscene_amd.samplers import VQDecoder
    g = load_golden('vqvae_' + tag)
    ch, ne = [int(v) for v in g['cfg']]
    c = escfg.vqvae_conf(ch).model.params
    vq = VQVAE(dict(c.ddconfig), ne, c.embed_dim)
    synth.seeded_fill_(vq, prefix='vqvae_%s.' % tag)
    dec = VQDecoder(vq, torch.device('cuda'))
    sdf = dec.decode_no_quant(g['z'])
    assert tuple(sdf.shape[1:]) == (1, 64, 64, 64)
    e = _rel(sdf[:, :, ::4, ::4, ::4], g['sdf_sub'])
    print('vqvae %s decode: fp16-MFMA SDF vs fp32 reference golden: rel err %.3e' % (tag, e))
    assert e < 2e-2
    ea = abs(sdf.double().abs().sum().item() - g['sdf_abs'].item()) / g['sdf_abs'].item()
    assert ea < 5e-3
    for st in dec._plans.values():                     # scratch claim of the decode plans: NaN-poisoned, same bits
        assert st['plan'].poison_scratch() > 0
    assert torch.equal(dec.decode_no_quant(g['z']), sdf)


@pytest.mark.parametrize('typ,concat,gold', [('echolayout', False, None), ('echoscene', False, None), ('echoscene', True, None),
                                             ('echoscene', False, 'scene_e2e_O2_tiny'), ('echolayout', False, 'scene_e2e_norel_tiny')])
def test_sgdiff_api_end_to_end_vs_reference_golden(typ, concat, gold):
    """model.SGDiff.SGDiff(...).sample_box_and_shape on the GPU == the reference's own call (tiny widths);
    ``concat``: the config/full_concat_mp.yaml model family; ``scene_e2e_O2_tiny``: the smallest scene (one object + the scene node);
    ``scene_e2e_norel_tiny`` (round 6): ``layout_branch.relation_condition: false`` -- golden from the reference built that way."""
    import sys
    from model.SGDiff import SGDiff          # the drop-in import path eval_3dfront.py uses
    g = load_golden(gold or ('scene_e2e_concat_tiny' if concat else 'scene_e2e_tiny'))
    objs, triples = g['objs'], g['triples']
    O = objs.shape[0]
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
    opt = escfg.tiny_diff_opt('cuda', concat=concat)
    if gold == 'scene_e2e_norel_tiny':
        opt.layout_branch.relation_condition = False
    m = SGDiff(typ, opt, synth.VOCAB, replace_latent=False, with_changes=True, residual=True,
               gconv_pooling='avg', with_angles=True, clip=True, separated=False)
    synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), prefix='e2e.diff.')
    if typ == 'echoscene':
        synth.seeded_fill_(m.diff.ShapeDiff.df, prefix='e2e.shape_df.')
        synth.seeded_fill_(m.diff.ShapeDiff.vqvae, prefix='e2e.vqvae.')
        m.diff.ShapeDiff.ddim_steps = 4
    m.diff.optimizer_ini()
    m.cuda()
    m.eval()
    kw = dict(layout_noise=synth.layout_noise(O, 8, 100, seed=7))
    if typ == 'echoscene':
        kw['shape_noise'] = synth.shape_noise(seed=7)
    d = m.sample_box_and_shape(objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda(), gen_shape=(typ == 'echoscene'), **kw)
    for k in ('sizes', 'translations', 'angles'):
        assert d[k].is_cuda and d[k].is_contiguous() and d[k].dtype == torch.float32
        assert _rel(d[k], g['%s_%s' % (typ, k)]) < 1e-4, k
    assert tuple(d['sizes'].shape) == (O, 3) and tuple(d['angles'].shape) == (O, 2)
    if typ == 'echoscene':
        assert tuple(d['shapes'].shape) == (O, 1, 64, 64, 64)
        # The decode starts with a nearest-codebook argmin (quantizer.py:80-84): a latent that differs in the 4th
        # digit (fp16 MFMA path) can flip a near-tie to another code, which changes the SDF locally by O(1).  So the
        # SDF is compared as a distribution: almost every sample within the fp16 tolerance, few flipped neighbourhoods.
        got, ref = d['shapes'][:, :, ::4, ::4, ::4].cpu(), g['echoscene_shapes']
        scale = ref.abs().max().item()
        bad = ((got - ref).abs() > 2e-2 * scale).float().mean().item()
        med = (got - ref).abs().median().item() / scale
        print('e2e echoscene SDF vs reference: %.3f%% of samples outside 2e-2, median rel err %.2e' % (100 * bad, med))
        # (two objects: a handful of flipped codes of the tiny 64-entry codebook already are 5.9 % of the 8192 samples)
        assert bad < (0.03 if O >= 8 else 0.10) and med < 2e-3
    else:
        assert 'shapes' not in d


def _build_sgdiff(typ, concat, opt=None, **flags):
    from model.SGDiff import SGDiff
    kw = dict(replace_latent=False, with_changes=True, residual=True, gconv_pooling='avg', with_angles=True, clip=True, separated=False)
    kw.update(flags)
    m = SGDiff(typ, opt or escfg.tiny_diff_opt('cuda', concat=concat), synth.VOCAB, **kw)
    synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), prefix='e2e.diff.')
    if typ == 'echoscene':
        synth.seeded_fill_(m.diff.ShapeDiff.df, prefix='e2e.shape_df.')
        synth.seeded_fill_(m.diff.ShapeDiff.vqvae, prefix='e2e.vqvae.')
        m.diff.ShapeDiff.ddim_steps = 4
    m.diff.optimizer_ini()
    m.cuda()
    m.eval()
    return m


def _check_sdf(got_full, ref_sub, what, max_bad=0.03):
    got, ref = got_full[:, :, ::4, ::4, ::4].cpu(), ref_sub
    scale = ref.abs().max().item()
    bad = ((got - ref).abs() > 2e-2 * scale).float().mean().item()
    med = (got - ref).abs().median().item() / scale
    print('%s SDF vs reference: %.3f%% of samples outside 2e-2, median rel err %.2e' % (what, 100 * bad, med))
    assert bad < max_bad and med < 2e-3


@pytest.mark.parametrize('fam,typ,concat', [('lay', 'echolayout', False), ('sc', 'echoscene', False), ('cat', 'echoscene', True)])
def test_sgdiff_editing_vs_reference_golden(fam, typ, concat):
    """sample_boxes_and_shape_with_changes / _with_additions vs the reference's own calls
    (EchoScene.py:422-532, EchoLayout.py:309-401; goldens: tests/golden/make_golden.py case_scene_edit): numpy RNG seeded on
    both sides for the 64-d change noise, unsorted ``manipulated_nodes``, two missing nodes.  Compared: the keep mask
    (type, shape, values exactly), the manipulator's conditioning (LayoutDiff.rel, 1e-4), boxes (1e-4), SDFs (fp16 path)."""
    g = load_golden('scene_edit_tiny')
    objs, triples = g['objs'], g['triples']
    O = objs.shape[0]
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
    manipulated, missing = [int(v) for v in g['manipulated']], [int(v) for v in g['missing']]
    m = _build_sgdiff(typ, concat)
    kw = dict(layout_noise=synth.layout_noise(O, 8, 100, seed=7))
    if typ == 'echoscene':
        kw.update(shape_noise=synth.shape_noise(seed=7), gen_shape=True)
    dec = (objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda())
    # ---- changes
    np.random.seed(123)
    keep, d = m.sample_boxes_and_shape_with_changes(*dec, *dec, manipulated, **kw)
    assert torch.is_tensor(keep) and keep.is_cuda and keep.dtype == torch.float32 and tuple(keep.shape) == (O, 1)
    assert torch.equal(keep.cpu(), g[fam + '_chg_keep'])
    assert _rel(m.diff.LayoutDiff.rel, g[fam + '_chg_rel']) < 1e-4
    for k in ('sizes', 'translations', 'angles'):
        assert _rel(d[k], g['%s_chg_%s' % (fam, k)]) < 1e-4, k
    if typ == 'echoscene':
        _check_sdf(d['shapes'], g[fam + '_chg_shapes'], fam + ' with_changes')
    # ---- additions (the encoder sees the graph without the added nodes)
    added = [mi + i for i, mi in enumerate(missing)]
    eo, et, keep_idx, keep_tri = synth.remove_nodes(objs, triples, added)
    enc = (eo.cuda(), et.cuda(), tf[keep_idx].cuda(), rf[keep_tri].cuda())
    np.random.seed(321)
    r = m.sample_boxes_and_shape_with_additions(*enc, *dec, missing, **kw)
    rel = m.diff.LayoutDiff.rel.clone()
    if typ == 'echolayout':
        d = r                                   # the facade drops keep here (SGDiff.py:113-115)
        assert isinstance(d, dict)
    else:
        keep, d = r
        assert torch.is_tensor(keep) and keep.is_cuda and tuple(keep.shape) == (O, 1)
        assert torch.equal(keep.cpu(), g[fam + '_add_keep'])
    assert _rel(rel, g[fam + '_add_rel']) < 1e-4
    for k in ('sizes', 'translations', 'angles'):
        assert _rel(d[k], g['%s_add_%s' % (fam, k)]) < 1e-4, k
    if typ == 'echoscene':
        _check_sdf(d['shapes'], g[fam + '_add_shapes'], fam + ' with_additions')
    else:
        # the model-level call returns keep as a python list (EchoLayout.py:394-401)
        k2, _ = m.diff.sampleBoxes_with_additions(*enc, *dec, missing, layout_noise=kw['layout_noise'])
        assert isinstance(k2, list) and k2 == [0 if i in added else 1 for i in range(O)]


@pytest.mark.parametrize('typ', ['echoscene', 'echolayout'])
def test_sgdiff_flag_matrix_vs_reference_golden(typ):
    """The other corner of the constructor's flag matrix (SGDiff.py:8-30; golden: make_golden.py case_scene_flags, the reference's own
    API): clip=False (128-d node vectors, denoiser_kwargs.using_clip False as eval_3dfront.py:386 sets it), residual=False (setup
    GCNs without skip projections), replace_latent=True (editing takes every latent from the manipulator, EchoScene.py:440-448) and a
    layout denoiser WITHOUT enable_t_emb (config/full.yaml / box.yaml carry no such key).  Plain sampling and with_changes."""
    g = load_golden('scene_flags_tiny')
    objs, triples = g['objs'], g['triples']
    O = objs.shape[0]
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
    opt = escfg.tiny_diff_opt('cuda')
    del opt.layout_branch.denoiser_kwargs['enable_t_emb']
    opt.layout_branch.denoiser_kwargs.using_clip = False
    m = _build_sgdiff(typ, False, opt=opt, clip=False, residual=False, replace_latent=True)
    assert not m.diff.LayoutDiff.df.model.enable_t_emb
    kw = dict(layout_noise=synth.layout_noise(O, 8, 100, seed=7))
    if typ == 'echoscene':
        kw.update(shape_noise=synth.shape_noise(seed=7), gen_shape=True)
    dec = (objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda())
    d = m.sample_box_and_shape(*dec, **kw)
    for k in ('sizes', 'translations', 'angles'):
        assert _rel(d[k], g['%s_%s' % (typ, k)]) < 1e-4, k
    if typ == 'echoscene':
        # the latents BEFORE the codebook argmin (the golden records what the reference's DDIM loop handed to its VQ-VAE): the
        # continuous quantity, at the fp16-operand tolerance.  The SDF after the argmin of a 64-entry random codebook is a
        # distribution statement (a 4th-digit difference flips near-ties: 1.5-3.4 % of the samples move with the summation order).
        ez = _rel(m.diff.ShapeDiff.gen_z, g['echoscene_z'])
        print('flags: latents after 4 DDIM steps vs reference: rel err %.2e' % ez)
        assert ez < 2e-2
        _check_sdf(d['shapes'], g['echoscene_shapes'], 'flags: plain', max_bad=0.05)
    manipulated = [int(v) for v in g['manipulated']]
    np.random.seed(123)
    keep, d = m.sample_boxes_and_shape_with_changes(*dec, *dec, manipulated, **kw)
    assert torch.equal(keep.cpu(), g[typ + '_chg_keep'])
    assert _rel(m.diff.LayoutDiff.rel, g[typ + '_chg_rel']) < 1e-4
    for k in ('sizes', 'translations', 'angles'):
        assert _rel(d[k], g['%s_chg_%s' % (typ, k)]) < 1e-4, k
    if typ == 'echoscene':
        assert _rel(m.diff.ShapeDiff.gen_z, g['echoscene_chg_z']) < 2e-2
        _check_sdf(d['shapes'], g['echoscene_chg_shapes'], 'flags: with_changes', max_bad=0.05)


def test_sgdiff_editing_index_edge_cases():
    """duplicates and out-of-range entries of manipulated_nodes behave like the reference's ``i in list`` tests:
    one draw per distinct in-range node, the others are ignored (no IndexError)."""
    O = 6
    objs, triples = synth.synthetic_graph(O, seed=4)
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=4)
    m = _build_sgdiff('echolayout', False)
    a = (objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda())
    noise = synth.layout_noise(O, 8, 100, seed=3)
    np.random.seed(5)
    keep1, d1 = m.sample_boxes_and_shape_with_changes(*a, *a, [3, 1], layout_noise=noise)
    rel1 = m.diff.LayoutDiff.rel.clone()
    np.random.seed(5)
    keep2, d2 = m.sample_boxes_and_shape_with_changes(*a, *a, [1, 3, 3, 17], layout_noise=noise)
    assert torch.equal(keep1, keep2) and keep1.flatten().tolist() == [1, 0, 1, 0, 1, 1]
    assert torch.equal(rel1, m.diff.LayoutDiff.rel)
    assert torch.equal(d1['sizes'], d2['sizes'])


def test_box_postprocess_matches_reference_helpers():
    """descale_box_params / postprocess_sincos2arctan on the device vs the oracle
    restatement of helpers/util.py:542-568; the box tensor is updated in place like the reference does."""
    from echoscene_amd.postprocess import descale_box_params, postprocess_sincos2arctan
    from oracle import echoscene_oracle as orc
    rs = np.random.RandomState(0)
    boxes = torch.from_numpy(rs.uniform(-1.2, 1.2, (33, 6)).astype(np.float32))
    sc = torch.from_numpy(rs.standard_normal((33, 2)).astype(np.float32))
    stats = np.concatenate([rs.uniform(0.1, 0.5, 3), rs.uniform(1.0, 3.0, 3), rs.uniform(-4, -2, 3), rs.uniform(2, 4, 3),
                            [-np.pi, np.pi]])
    b = boxes.cuda()
    r = descale_box_params(b, stats=stats)
    assert r.data_ptr() == b.data_ptr()
    assert torch.allclose(b.cpu(), orc.descale_box_params(boxes, stats), atol=1e-6, rtol=1e-6)
    g = load_golden('box_post')                                     # the reference's own outputs, incl. angle=True (7 columns)
    b6 = g['boxes'].cuda()
    descale_box_params(b6, stats=g['stats'].numpy())
    assert torch.allclose(b6.cpu(), g['boxes_out'], atol=1e-6, rtol=1e-6)
    b7 = g['boxes7'].cuda()
    assert descale_box_params(b7, stats=g['stats'].numpy(), angle=True).data_ptr() == b7.data_ptr()
    assert torch.allclose(b7.cpu(), g['boxes7_out'], atol=1e-6, rtol=1e-6)
    a = postprocess_sincos2arctan(sc.cuda())
    assert tuple(a.shape) == (33, 1)
    assert torch.allclose(a.cpu(), orc.sincos2arctan(sc), atol=2e-6)


@pytest.mark.parametrize('B,n,m', [(1, 5000, 5000), (3, 700, 1234), (2, 1, 9)])
def test_chamfer_vs_oracle(B, n, m):
    """The reference's only native kernel (extension/old_chamfer), re-designed for CDNA4: squared NN distances,
    indices and the autograd backward vs the CPU oracle (shapes incl. consistency_check.py's 1x5000x3 clouds)."""
    from echoscene_amd.chamfer import chamferDist
    from oracle import echoscene_oracle as orc
    rs = np.random.RandomState(n)
    a = torch.from_numpy(rs.standard_normal((B, n, 3)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal((B, m, 3)).astype(np.float32))
    ac, bc = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    d1, d2 = chamferDist()(ac, bc)
    r1, i1, r2, i2 = orc.chamfer_forward(a, b)
    assert torch.allclose(d1.detach().cpu(), r1, atol=1e-6, rtol=1e-5) and torch.allclose(d2.detach().cpu(), r2, atol=1e-6, rtol=1e-5)
    w1 = torch.from_numpy(rs.standard_normal((B, n)).astype(np.float32))
    w2 = torch.from_numpy(rs.standard_normal((B, m)).astype(np.float32))
    ((d1 * w1.cuda()).sum() + (d2 * w2.cuda()).sum()).backward()
    g1, g2 = orc.chamfer_backward(a, b, w1, w2, i1, i2)
    assert torch.allclose(ac.grad.cpu(), g1, atol=2e-4, rtol=1e-4) and torch.allclose(bc.grad.cpu(), g2, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('O', [2, 5])
def test_small_and_odd_graphs_vs_oracle(O):
    """Edge sizes: the smallest scene graph (one object + the scene node, a single triple) and an odd object count --
    both loops against the CPU oracle (tiny widths)."""
    from echoscene_amd.model.unet import UNet1DModel, DiffusionUNet
    from echoscene_amd.samplers import LayoutDenoiser, ShapeDenoiser
    from oracle import echoscene_oracle as orc
    dev = torch.device('cuda')
    objs, triples = synth.synthetic_graph(O, seed=21)
    assert triples.shape[0] >= 1
    kw = dict(escfg.layout_denoiser_kwargs(128))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='edge.layout.')
    oe = torch.randn(O, 640, generator=torch.Generator().manual_seed(3))
    noise = synth.layout_noise(O, 8, 100, seed=11)
    x = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(100), dev).sample(oe, triples, noise, n_steps=12).cpu()
    ref = orc.layout_sample_loop({k: v.detach() for k, v in net.state_dict().items()}, oe, triples, noise, 100, n_steps=12)
    assert (x - ref).abs().max().item() < 2e-4
    p = escfg.shape_unet_params(32)
    p['context_dim'] = 64
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix='edge.shape.')
    uc = torch.randn(O, 1, 64, generator=torch.Generator().manual_seed(4))
    z = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev).sample(uc, triples, synth.shape_noise(seed=7)).cpu()
    zref = orc.shape_sample_loop({k[len('diffusion_net.'):]: v.detach() for k, v in df.state_dict().items()}, uc, triples,
                                 synth.shape_noise(seed=7), S=4)
    assert _rel(z, zref) < 2e-2


def test_collated_batch_equals_single_scenes():
    """Batch mode (bench.py --scaling weak, BASELINE configs[4]): scenes collated into one block-diagonal graph as the
    reference's collate_fn does give, per scene, the result of sampling that scene alone (no cross-scene edges)."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    dev = torch.device('cuda')
    p = escfg.shape_unet_params(32)
    p['context_dim'] = 64
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix='batch.shape.')
    graphs = [synth.synthetic_graph(n, seed=30 + i) for i, n in enumerate((4, 3))]
    ucs = [torch.randn(n, 1, 64, generator=torch.Generator().manual_seed(40 + i)) for i, n in enumerate((4, 3))]
    noise1 = synth.shape_noise(seed=7)
    den = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev)
    singles = [den.sample(u, g[1], noise1).cpu() for g, u in zip(graphs, ucs)]
    _, tri_all = synth.collate_graphs(graphs)
    zb = den.sample(torch.cat(ucs), tri_all, noise1).cpu()
    assert _rel(zb, torch.cat(singles)) < 2e-3      # tile sizes depend on the object count: fp16-operand rounding noise


def test_full_size_permutation_equivariance():
    """Size-independent property at BASELINE.json's full size (32-node graph, shipped widths): relabelling the objects
    (rows of x / obj_embed permuted, triple endpoints renamed, triple order kept) permutes the outputs of both
    denoisers -- exercises the gather / segmented-mean indexing and the per-object independence of every volume kernel
    at O = 32, where no CPU oracle run is affordable."""
    from echoscene_amd.model.unet import UNet1DModel, DiffusionUNet
    from echoscene_amd.samplers import LayoutDenoiser, ShapeDenoiser
    dev = torch.device('cuda')
    O = 32
    objs, triples = synth.synthetic_graph(O, seed=100)
    gen = torch.Generator().manual_seed(77)
    perm = torch.randperm(O, generator=gen)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(O)
    tri_p = triples.clone()
    tri_p[:, 0], tri_p[:, 2] = inv[triples[:, 0]], inv[triples[:, 2]]
    # layout denoiser, full width
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(512))
    synth.seeded_fill_(net, prefix='perm.layout.')
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    x = torch.randn(O, 8, generator=gen)
    oe = torch.randn(O, 640, generator=gen)
    e0 = den.eps(x, oe, triples, iteration=400).cpu()
    e1 = den.eps(x[perm], oe[perm], tri_p, iteration=400).cpu()
    assert torch.isfinite(e0).all() and (e1 - e0[perm]).abs().max().item() < 1e-5 * max(1.0, e0.abs().max().item())
    del den, net
    # shape denoiser, full width
    conf = escfg.shape_df_conf(224)
    df = DiffusionUNet(conf.unet.params, conditioning_key='crossattn')
    synth.seeded_fill_(df, prefix='perm.shape.')
    sden = ShapeDenoiser(df, conf.model.params, ddim_steps=100, device=dev)
    xs = torch.randn(O, 3, 16, 16, 16, generator=gen)
    uc = torch.randn(O, 1, 1280, generator=gen)
    s0 = sden.eps(xs, uc, triples, iteration=40).cpu()
    s1 = sden.eps(xs[perm], uc[perm], tri_p, iteration=40).cpu()
    assert torch.isfinite(s0).all()
    assert _rel(s1, s0[perm]) < 1e-5, 'objects are not independent / indexing depends on the labelling'
    s2 = sden.eps(xs, uc, triples, iteration=40).cpu()
    assert torch.equal(s0, s2), 'the full-size step is not deterministic'


def test_chamfer_exact_lattice_fixtures():
    """SURVEY 8(f4) pin: integer-exact fixtures (tests/golden/make_chamfer_lattice.py) -- every
    dx*dx+dy*dy+dz*dz is exact in fp32, so dist / idx / grads must be BIT-identical to the int64 ground truth, including
    the first-minimum rule across the reference's 512-point tile seam (chamfer.cu:14-16,118-131) and this kernel's
    2048-point LDS tile; n, m are not multiples of 256 / 2048."""
    import ctypes as C
    import os
    from conftest import GOLDEN
    from echoscene_amd import hip
    from echoscene_amd.chamfer import chamferDist
    d = np.load(os.path.join(GOLDEN, 'chamfer_lattice.npz'))
    sc = float(d['scale'])
    for tag in 'abc':
        x1 = torch.from_numpy(d[tag + '_xyz1'].astype(np.float32) * sc).cuda()
        x2 = torch.from_numpy(d[tag + '_xyz2'].astype(np.float32) * sc).cuda()
        B, n, m = x1.shape[0], x1.shape[1], x2.shape[1]
        d1, d2 = torch.zeros(B, n, device='cuda'), torch.zeros(B, m, device='cuda')
        i1 = torch.zeros(B, n, dtype=torch.int32, device='cuda')
        i2 = torch.zeros(B, m, dtype=torch.int32, device='cuda')
        p = lambda t: C.c_void_p(t.data_ptr())
        hip.check(hip.lib().es_chamfer_forward(p(x1), p(x2), B, n, m, p(d1), p(i1), p(d2), p(i2), hip.current_stream()),
                  'es_chamfer_forward')
        torch.cuda.synchronize()
        assert torch.equal(d1.cpu(), torch.from_numpy(d[tag + '_dist1'])), tag
        assert torch.equal(d2.cpu(), torch.from_numpy(d[tag + '_dist2'])), tag
        assert torch.equal(i1.cpu(), torch.from_numpy(d[tag + '_idx1'])), tag
        assert torch.equal(i2.cpu(), torch.from_numpy(d[tag + '_idx2'])), tag
        if tag == 'a':
            assert int(i1[0, 3]) == 100          # the planted duplicate: first copy wins over 511/512/700/2047/2048/2100
        # backward through the autograd wrapper the reference's consistency_check.py uses
        a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        o1, o2 = chamferDist()(a, b)
        g1 = torch.from_numpy(d[tag + '_g1'].astype(np.float32) * 0.25).cuda()
        g2 = torch.from_numpy(d[tag + '_g2'].astype(np.float32) * 0.25).cuda()
        torch.autograd.backward([o1, o2], [g1, g2])
        assert torch.equal(a.grad.cpu(), torch.from_numpy(d[tag + '_grad1'])), tag
        assert torch.equal(b.grad.cpu(), torch.from_numpy(d[tag + '_grad2'])), tag


def test_configs4_rank_shapes_on_one_gpu():
    """BASELINE configs[4] as ONE rank of the 8-GPU batch run sees it (bench.py --scaling weak --scenes-per-gpu 8), executed on one
    GPU at the shipped widths: (a) the layout denoiser on a 64-scene collated batch (2048 node rows, ~8000 triple rows: 64
    row tiles per rows-kernel launch) and (b) the shape denoiser on 8 collated scenes = 256 objects (the multi-scene plan,
    >= 2048 tiles per conv launch).  Checked through the size-independent property the collate gives: every scene of the batch
    == that scene sampled alone (no cross-scene edges); exact for the fp32 rows path, fp16-tolerance for the volume path
    (tile / split-K choices depend on the object count)."""
    from echoscene_amd.model.unet import UNet1DModel, DiffusionUNet
    from echoscene_amd.samplers import LayoutDenoiser, ShapeDenoiser
    dev = torch.device('cuda')
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip('needs ~70 GB of HBM for the 256-object plan')
    O = 32
    # ---- (a) layout: 64 scenes x 32 nodes
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(512))
    synth.seeded_fill_(net, prefix='c4.layout.')
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    graphs = [synth.synthetic_graph(O, seed=200 + s) for s in range(64)]
    oes = [torch.randn(O, 640, generator=torch.Generator().manual_seed(300 + s)) for s in range(64)]
    xs = [torch.randn(O, 8, generator=torch.Generator().manual_seed(400 + s)) for s in range(64)]
    _, tri_all = synth.collate_graphs(graphs)
    assert tri_all.shape[0] > 7000
    eps_b = den.eps(torch.cat(xs), torch.cat(oes), tri_all, iteration=123).cpu()
    assert torch.isfinite(eps_b).all()
    for sidx in (0, 37, 63):
        e1 = den.eps(xs[sidx], oes[sidx], graphs[sidx][1], iteration=123).cpu()
        assert torch.equal(eps_b[sidx * O:(sidx + 1) * O], e1), sidx          # per-row arithmetic is independent of the batch
    del den
    # ---- (b) shape: 8 scenes x 32 objects, two DDIM steps
    conf = escfg.shape_df_conf(224)
    df = DiffusionUNet(conf.unet.params, conditioning_key='crossattn')
    synth.seeded_fill_(df, prefix='c4.shape.')
    sden = ShapeDenoiser(df, conf.model.params, ddim_steps=100, device=dev)
    graphs8 = graphs[:8]
    ucs = [torch.randn(O, 1, 1280, generator=torch.Generator().manual_seed(500 + s)) for s in range(8)]
    noise1 = synth.shape_noise(seed=7)
    _, tri8 = synth.collate_graphs(graphs8)
    zb = sden.sample(torch.cat(ucs), tri8, noise1, n_steps=2).cpu()
    assert tuple(zb.shape) == (8 * O, 3, 16, 16, 16) and torch.isfinite(zb).all()
    for sidx in (0, 7):
        z1 = sden.sample(ucs[sidx], graphs8[sidx][1], noise1, n_steps=2).cpu()
        e = _rel(zb[sidx * O:(sidx + 1) * O], z1)
        print('configs[4] rank shapes: scene %d of the 8-scene batch vs alone: rel err %.2e' % (sidx, e))
        assert e < 2e-3


def test_model_files_replayed_by_a_c_host(tmp_path):
    """SURVEY.md section 8(b) / VERDICT r2 #9: the coarse C entry points.  The planner writes model files (plan + buffers) for the
    three loops of a tiny scene; a C program (tests/c/replay_model.c, compiled here with hipcc against the library, no Python, no
    torch) loads them and runs es_layout_sample / es_shape_sample / es_vq_decode; results == the Python host's, bit for bit."""
    import os
    import subprocess
    from echoscene_amd.model.unet import UNet1DModel, DiffusionUNet
    from echoscene_amd.model.vqvae import VQVAE
    from echoscene_amd.samplers import LayoutDenoiser, ShapeDenoiser, VQDecoder
    from echoscene_amd import hip
    dev = torch.device('cuda')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'replay_model')
    libdir = os.path.dirname(hip.LIB_PATH)
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    subprocess.check_call(['gcc', os.path.join(root, 'tests', 'c', 'replay_model.c'), '-I', os.path.join(root, 'include'),
                           '-I', rocm + '/include', '-D__HIP_PLATFORM_AMD__', '-L', libdir, '-lechoscene_hip', '-L', rocm + '/lib',
                           '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,' + rocm + '/lib', '-o', exe])
    O = 6
    objs, triples = synth.synthetic_graph(O, seed=21)

    def run_c(kind, model, arr, n_steps, shape):
        fi, fo = str(tmp_path / (kind + '_in.f32')), str(tmp_path / (kind + '_out.f32'))
        arr.detach().cpu().contiguous().numpy().astype(np.float32).tofile(fi)
        r = subprocess.run([exe, kind, model, fi, fo, str(n_steps)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        return torch.from_numpy(np.fromfile(fo, dtype=np.float32).reshape(shape))

    # layout loop
    kw = dict(escfg.layout_denoiser_kwargs(128))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='cmodel.layout.')
    oe = torch.randn(O, 640, generator=torch.Generator().manual_seed(2))
    noise = synth.layout_noise(O, 8, 100, seed=7)
    lden = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(100), dev)
    x_py = lden.sample(oe, triples, noise, n_steps=30).cpu()
    f_lay = str(tmp_path / 'layout.esm')
    lden.save_model(f_lay, oe, triples)
    # round 5: the file records the route options it was built under (es_model_load refuses a file written under other values)
    import ctypes
    from conftest import route_options
    from echoscene_amd import hip
    rec = ctypes.create_string_buffer(1024)
    assert hip.lib().es_model_file_options(f_lay.encode(), rec, 1024) == 0
    assert dict(kv.split('=') for kv in rec.value.decode().strip(';').split(';')) == route_options()
    hip.check(hip.lib().es_vol_set_option(b'conv_wss_target', 512), 'es_vol_set_option')
    try:
        assert not hip.lib().es_model_load(f_lay.encode()), 'a model file written under other route options must be refused'
        assert b'route options' in hip.lib().es_last_error()
    finally:
        hip.check(hip.lib().es_vol_set_option(b'conv_wss_target', 256), 'es_vol_set_option')
    x_c = run_c('layout', f_lay, noise.reshape(noise.shape[0], -1), 30, (O, 8))
    assert torch.equal(x_c, x_py), (x_c - x_py).abs().max()
    # shape loop
    p = escfg.shape_unet_params(32)
    p['context_dim'] = 64
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix='cmodel.shape.')
    uc = torch.randn(O, 1, 64, generator=torch.Generator().manual_seed(3))
    sden = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev)
    noise1 = synth.shape_noise(seed=7)
    z_py = sden.sample(uc, triples, noise1).cpu()
    f_shp = str(tmp_path / 'shape.esm')
    sden.save_model(f_shp, uc, triples)
    z_c = run_c('shape', f_shp, noise1.expand(O, 3, 16, 16, 16), 4, (O, 3, 16, 16, 16))
    assert torch.equal(z_c, z_py), (z_c - z_py).abs().max()
    # VQ-VAE decode
    c = escfg.vqvae_conf(32).model.params
    vq = VQVAE(dict(c.ddconfig), 64, c.embed_dim)
    synth.seeded_fill_(vq, prefix='cmodel.vq.')
    dec = VQDecoder(vq, dev, chunk=2)
    zz = z_py[:2]
    sdf_py = dec.decode_no_quant(zz).cpu()
    f_vq = str(tmp_path / 'vq.esm')
    dec.save_model(f_vq, 2)
    sdf_c = run_c('vq', f_vq, zz, 0, tuple(sdf_py.shape))
    assert torch.equal(sdf_c, sdf_py), (sdf_c - sdf_py).abs().max()


def test_sharded_step_is_one_captured_graph_with_rccl_exchange():
    """SURVEY.md section 8(e) / VERDICT r2 #6: a sharded DDIM step = this rank's stem ops -> RCCL all-gather of the codes ->
    everything else, captured as ONE graph (samplers.ShapeDenoiser.step_graph).  One-GPU form: a 1-rank NCCL group and the
    sharded step structure forced at world == 1 (tools/probe_step_graph.py, own process: it owns a process group); the captured loop
    must reproduce the ordinary single-graph run bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'probe_step_graph.py')], cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert 'STEP_GRAPH_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_bench_two_ranks_on_one_gpu_strong_and_weak():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with two ranks
    sharing the one GPU of the test box over gloo (ES_DIST_BACKEND=gloo: NCCL refuses two ranks on one device; the exchange is
    staged through host memory in that mode, everything else is the N > 1 code path: object sharding, stem graph -> all-gather
    -> main graph with the layout step as a parallel branch, max-over-ranks timing, one JSON line from rank 0)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra, scaling in ((['--nodes', '8'], 'strong'), (['--scaling', 'weak', '--scenes-per-gpu', '1', '--nodes', '8'], 'weak')):
        with socket.socket() as sck:
            sck.bind(('127.0.0.1', 0))
            port = sck.getsockname()[1]
        env = dict(os.environ, ES_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
               '--no-cpu-baseline', '--no-sub-records'] + extra
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, r.stdout[-1500:]
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['scaling'] == scaling and d['steps'] == 3 and d['value'] > 0
        assert d['config']['scenes'] == (2 if scaling == 'weak' else 1)
        assert 'roofline' in d and d['roofline']['achieved'] > 0
    # the sharded run must reproduce the single-rank run: same seeded latents, deterministic shards -> identical final latents
    crcs = []
    for nproc in (1, 2):
        with socket.socket() as sck:
            sck.bind(('127.0.0.1', 0))
            port = sck.getsockname()[1]
        env = dict(os.environ, ES_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(nproc), '--steps', '3', '--warmup', '1',
               '--no-cpu-baseline', '--no-sub-records', '--nodes', '8', '--check', '--deterministic']      # the canonical arithmetic: bit-exact across world sizes
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
        assert d['check']['objects'] == 8
        crcs.append((d['check']['latents_crc32'], d['check']['abs_sum']))
    assert crcs[0] == crcs[1], 'the 2-rank sharded run does not reproduce the 1-rank latents: %s' % (crcs,)
    # the plain-python form of the contract (`python bench.py --gpus N`, no rendezvous environment): bench.py starts the ranks itself
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['ES_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
           '--no-sub-records', '--nodes', '8', '--check', '--deterministic']
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and (d['check']['latents_crc32'], d['check']['abs_sum']) == crcs[0]
