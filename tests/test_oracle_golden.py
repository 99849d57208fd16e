"""Pins the CPU oracle (oracle/echoscene_oracle.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  Weights are regenerated on this side from
the seeded rule, through this build's own parameter-holder trees -- so these tests also pin
the holders' state_dict key names and shapes to the reference's (any mismatch would change
the seeded values and fail numerically).

Tolerances: same arithmetic (ATen fp32 on CPU) on both sides, only op grouping differs
-> 2e-5 abs on O(1) values; the tables must be bit-identical.
"""
import pytest
import torch

from conftest import load_golden, seeded_state_dict
from echoscene_amd import synth, config as escfg
from echoscene_amd.model.graph import GraphTripleConvNet
from echoscene_amd.model.unet import UNet1DModel, DiffusionUNet
from oracle import echoscene_oracle as orc


def _close(a, b, atol, rtol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), 'max abs err %.3e (ref scale %.3e)' % (
        err, b.abs().max().item())


@pytest.mark.parametrize('tag', ['res_bn', 'plain', 'ragged', 'one_node'])
def test_gcn(tag):
    """('ragged': a node without triples, a hub, a repeated pair, a self-loop; 'one_node': one node, one self-loop triple -- both with the
    weights of 'res_bn')"""
    g = load_golden('gcn_' + tag)
    din, dp, nl, H, res, bn, dout = [int(v) for v in g['cfg']]
    net = GraphTripleConvNet(din, dp, num_layers=nl, hidden_dim=H, residual=bool(res),
                             mlp_normalization='batch' if bn else 'none', output_dim=dout)
    sd = seeded_state_dict(net, 'gcn_%s.' % (tag if tag in ('res_bn', 'plain') else 'res_bn'))
    tri = g['triples']
    edges = torch.stack([tri[:, 0], tri[:, 2]], 1)
    o, p = orc.gcn_net({'n.' + k: v for k, v in sd.items()}, 'n', g['obj'], g['pred'], edges)
    _close(o, g['out_obj'], 1e-5)
    _close(p, g['out_pred'], 1e-5)


@pytest.mark.parametrize('tag', ['sum_g8', 'sum_ragged', 'wavg_g8', 'wavg_ragged'])
def test_gcn_other_poolings(tag):
    """pooling='sum' and the learned pooling='wAvg' (WeightNetGCN, model/graph.py:37-86, 163-184) against the reference's own
    GraphTripleConvNet; 'ragged' holds a node without triples (wAvg: 0 / (0 + 1e-4))."""
    g = load_golden('gcn_' + tag)
    din, dp, nl, H, res, bn, dout, code = [int(v) for v in g['cfg']]
    pool = ('avg', 'sum', 'wAvg')[code]
    net = GraphTripleConvNet(din, dp, num_layers=nl, hidden_dim=H, residual=bool(res), pooling=pool,
                             mlp_normalization='batch' if bn else 'none', output_dim=dout)
    sd = seeded_state_dict(net, 'gcn_%s.' % pool)
    if pool == 'wAvg':
        assert 'gconvs.0.weightNet.Net_s.0.weight' in sd and 'gconvs.2.weightNet.down_sample_pred.bias' in sd
    tri = g['triples']
    edges = torch.stack([tri[:, 0], tri[:, 2]], 1)
    o, p = orc.gcn_net({'n.' + k: v for k, v in sd.items()}, 'n', g['obj'], g['pred'], edges, pooling=pool)
    _close(o, g['out_obj'], 1e-5)
    _close(p, g['out_pred'], 1e-5)


def test_gcn_pooling_argument_is_checked():
    with pytest.raises(ValueError):
        GraphTripleConvNet(8, 8, pooling='max')


def _unet1d_sd(mc, ctx, prefix, t_emb=True):
    kw = dict(escfg.layout_denoiser_kwargs(mc))
    kw['concat_dim'] = kw['crossattn_dim'] = ctx
    if not t_emb:
        del kw['enable_t_emb']              # config/box.yaml, config/full.yaml: no key -> the constructor's default (False)
    return seeded_state_dict(UNet1DModel(**kw), prefix)


def test_unet1d_tiny():
    g = load_golden('unet1d_tiny')
    sd = _unet1d_sd(128, 128, 'unet1d_tiny.')
    eps = orc.unet1d_forward(sd, g['box'], g['obj_embed'], g['triples'], g['t'])
    _close(eps, g['eps'], 2e-5)


def test_ddpm_tables_bit_exact():
    g = load_golden('ddpm_tables_1000')
    tab = orc.ddpm_tables(1e-4, 0.02, 1000)
    for k, v in tab.items():
        if k in ('betas', 'posterior_variance'):        # (inputs of the 'fixedlarge' variance; pinned through sampler_variants_tiny)
            continue
        assert torch.equal(v, g[k]), k
    g100 = load_golden('layout_loop_tiny')
    tab = orc.ddpm_tables(1e-4, 0.02, 100)
    for k, v in tab.items():
        if k in ('betas', 'posterior_variance'):
            continue
        assert torch.equal(v, g100['tab100_' + k]), k


def test_layout_loop_tiny_100_steps():
    """BASELINE.json configs[0]: 8-node graph, 100 DDPM steps, injected noise."""
    g = load_golden('layout_loop_tiny')
    sd = _unet1d_sd(128, 128, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    x = orc.layout_sample_loop(sd, g['obj_embed'], g['triples'], noise, time_num=100)
    _close(x, g['x_final'], 1e-4)


@pytest.mark.slow
def test_unet1d_full():
    g = load_golden('unet1d_full')
    sd = _unet1d_sd(512, 1280, 'unet1d_full.')
    for O in (8, 32):
        t = torch.full((O,), 617, dtype=torch.int64)
        eps = orc.unet1d_forward(sd, g['box%d' % O], g['obj_embed%d' % O], g['triples%d' % O], t)
        _close(eps, g['eps%d' % O], 5e-5)
    noise = synth.layout_noise(8, 8, 1000, seed=7)[:11]
    tr = []
    x = orc.layout_sample_loop(sd, g['loop_obj_embed'], g['loop_triples'], noise, time_num=1000,
                               n_steps=10, trace=tr)
    _close(tr[0], g['loop_x1'], 5e-5)
    _close(x, g['loop_x10'], 1e-4)


def test_layout_loop_tiny_without_time_embedding():
    """config/box.yaml / full.yaml: the layout denoiser without ``enable_t_emb`` (denoise_net.py:505,735-740,766-768): the box GCN's
    node vectors are [obj_embed | box] only.  All 100 steps of the tiny loop vs the reference's own DiffusionPoint."""
    g = load_golden('layout_loop_tiny_no_temb')
    sd = _unet1d_sd(128, 128, 'unet1d_tiny_no_temb.', t_emb=False)
    assert not any(k.startswith('box_time_emb') for k in sd)
    x = orc.layout_sample_loop(sd, g['obj_embed'], g['triples'], synth.layout_noise(8, 8, 100, seed=7), time_num=100,
                               enable_t_emb=False)
    _close(x, g['x_final'], 1e-4)


@pytest.mark.slow
def test_unet1d_full_without_time_embedding():
    g = load_golden('unet1d_full_no_temb')
    sd = _unet1d_sd(512, 1280, 'unet1d_full_no_temb.', t_emb=False)
    for O in (8, 32):
        t = torch.full((O,), 617, dtype=torch.int64)
        eps = orc.unet1d_forward(sd, g['box%d' % O], g['obj_embed%d' % O], g['triples%d' % O], t, enable_t_emb=False)
        _close(eps, g['eps%d' % O], 5e-5)
    noise = synth.layout_noise(8, 8, 1000, seed=7)[:11]
    x = orc.layout_sample_loop(sd, g['loop_obj_embed'], g['loop_triples'], noise, time_num=1000, n_steps=10, enable_t_emb=False)
    _close(x, g['loop_x10'], 1e-4)


def test_sampler_options_clip_denoised_and_ddim_eta():
    """The sampler options beyond the shipped call: clip_denoised=True of the layout loop (diffusion_ddpm.py:243-244) and eta != 0 of the
    DDIM sampler (samplers/ddim.py:256-260) -- goldens from the reference's own loops (make_golden.py case_sampler_options); the sigma
    tables of the oracle AND of the product's ShapeSchedule must be bit-identical to the reference's."""
    from echoscene_amd.schedules import ShapeSchedule
    g = load_golden('sampler_options_tiny')
    sd = _unet1d_sd(128, 128, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    x = orc.layout_sample_loop(sd, g['layout_obj_embed'], g['layout_triples'], noise, time_num=100, clip_denoised=True)
    _close(x, g['layout_x_final_clip'], 1e-4)
    plain = load_golden('layout_loop_tiny')['x_final']
    assert (x - plain).abs().max() > 1e-3, 'the clamp must be active on this trajectory (else the golden pins nothing)'
    ac = orc.shape_alphas_cumprod()
    for S in (4, 100):
        assert torch.equal(orc.ddim_sigmas(ac, 0.7, S), g['ddim_sigmas_%d' % S].float())
        sch = ShapeSchedule(S, eta=0.7)
        assert torch.equal(sch.ddim_sigmas, g['ddim_sigmas_%d' % S].float())
        assert sch.coef.shape == (S, 5) and torch.equal(sch.coef[:, 4], sch.ddim_sigmas.flip(0))
        assert torch.equal(ShapeSchedule(S).coef, sch.coef[:, :4]) is False        # sqrt(1 - a_prev - sigma^2) differs from eta = 0
    dsd = _unet3d_sd(32, 64, 'unet3d_tiny.')
    step_noise = torch.stack([torch.from_numpy((__import__('numpy').random.RandomState(900 + k).standard_normal((4, 3, 16, 16, 16))).astype('float32'))
                              for k in range(4)])
    z = orc.shape_sample_loop(dsd, g['ddim_uc_s'], g['ddim_triples'], synth.shape_noise(seed=7), S=4, eta=0.7, step_noise=step_noise)
    _close(z, g['ddim_z_final_eta07'], 2e-4)


@pytest.mark.parametrize('tag,concat', [('crossattn', False), ('concat', True)])
def test_unet1d_model_channels_384(tag, concat):
    """GroupNorm32 groups that are not a power of two (model_channels = 384: 12 / 24 / 36 channels per group; reference: any
    channels % 32 == 0, ldm_diffusion_util.py:222-239): the oracle against the reference's eps and 3 loop steps."""
    g = load_golden('unet1d_mc384_' + tag)
    kw = dict(escfg.layout_denoiser_kwargs(384, concat=concat))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    sd = seeded_state_dict(UNet1DModel(**kw), 'unet1d_mc384_%s.' % tag)
    eps = orc.unet1d_forward(sd, g['box'], g['obj_embed'], g['triples'], g['t'])      # (the family is read off the input conv's width)
    _close(eps, g['eps'], 5e-5)
    if not concat:                                     # (one loop keeps the CPU suite short; the HIP test runs both)
        x = orc.layout_sample_loop(sd, g['loop_obj_embed'], g['loop_triples'], synth.layout_noise(8, 8, 100, seed=11), time_num=100,
                                   n_steps=3)
        _close(x, g['loop_x3'], 1e-4)


def test_sampler_variants_schedules_x0_prediction_fixedlarge():
    """The other parameterisations GaussianDiffusion samples with (get_betas warm-up schedules diffusion_ddpm.py:38-58, x0-prediction
    :246-254, 'fixedlarge' :224-235): goldens from the reference's own gen_samples_sg (make_golden.py case_sampler_variants).  The
    oracle restates the branches; the product expresses all of them as ONE coefficient table of the same update op
    (schedules.LayoutSchedule) -- both are checked: the oracle loop against the reference's results, the product's table against the
    oracle's step on random data, bit for bit."""
    import numpy as np
    from echoscene_amd.schedules import LayoutSchedule, layout_betas
    g = load_golden('sampler_variants_tiny')
    for st in ('warm0.1', 'warm0.2', 'warm0.5'):
        ref = g['betas_' + st.replace('.', '')].double().numpy()
        assert np.array_equal(orc.get_betas(st, 1e-4, 0.02, 1000), ref) and np.array_equal(layout_betas(st, 1e-4, 0.02, 1000), ref)
    assert float(g['cosine_raises']) == 1.0
    for fn in (orc.get_betas, layout_betas):
        with pytest.raises(UnboundLocalError):
            fn('cosine', 1e-4, 0.02, 1000)
        with pytest.raises(NotImplementedError):
            fn('quadratic', 1e-4, 0.02, 1000)
    sd = _unet1d_sd(128, 128, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    plain = load_golden('layout_loop_tiny')['x_final']
    cases = (('warm01_large', dict(schedule_type='warm0.1', model_var_type='fixedlarge'), False),
             ('warm05_x0', dict(schedule_type='warm0.5', model_mean_type='x0'), False),
             ('warm02_x0_large_clip', dict(schedule_type='warm0.2', model_mean_type='x0', model_var_type='fixedlarge'), True))
    for tag, kw, clip in cases:
        x = orc.layout_sample_loop(sd, g['obj_embed'], g['triples'], noise, time_num=100, clip_denoised=clip, **kw)
        _close(x, g['x_final_' + tag], 1e-4)
        assert (x - plain).abs().max() > 1e-3
        # the product's table: x0 = c0 x - c1 out; mean = c2 x0 + c3 x; x' = mean + c4 noise (k_ddpm_update, no contraction)
        sch = LayoutSchedule(100, 1e-4, 0.02, **kw)
        tab = orc.ddpm_tables(1e-4, 0.02, 100, kw['schedule_type'])
        rs = np.random.RandomState(5)
        for t in (99, 57, 10, 1, 0):
            xx, out, nz = (torch.from_numpy(rs.standard_normal((8, 8)).astype('float32')) for _ in range(3))
            want = orc.ddpm_step(tab, xx, out, t, nz, clip, kw.get('model_mean_type', 'eps'), kw.get('model_var_type', 'fixedsmall'))
            c = sch.coef[99 - t]
            x0 = c[0] * xx - c[1] * out
            if clip:
                x0 = torch.clamp(x0, -1.0, 1.0)
            got = (c[2] * x0 + c[3] * xx) + c[4] * nz
            assert torch.equal(got, want), (tag, t)
    with pytest.raises(NotImplementedError):
        LayoutSchedule(100, model_var_type='learned')
    with pytest.raises(NotImplementedError):
        LayoutSchedule(100, model_mean_type='v')


def _unet3d_sd(mc, ctx, prefix):
    p = escfg.shape_unet_params(mc)
    p['context_dim'] = ctx
    sd = seeded_state_dict(DiffusionUNet(p), prefix)
    return {k[len('diffusion_net.'):]: v for k, v in sd.items()}


def test_unet3d_tiny():
    g = load_golden('unet3d_tiny')
    sd = _unet3d_sd(32, 64, 'unet3d_tiny.')
    eps = orc.unet3d_forward(sd, g['x'], g['uc_s'], g['triples'], g['t'])
    _close(eps, g['eps'], 5e-5)


def test_ddim_schedule_and_loop_tiny():
    g = load_golden('ddim_tiny')
    ac = orc.shape_alphas_cumprod()
    assert torch.equal(ac, g['alphas_cumprod'])
    ts, a, ap, s1m = orc.ddim_schedule(ac, 4)
    assert (ts == g['ddim_timesteps'].numpy()).all()
    assert torch.equal(a, g['ddim_alphas'])
    assert torch.equal(ap, g['ddim_alphas_prev'].float())
    assert torch.equal(s1m, g['ddim_sqrt_one_minus_alphas'])
    g100 = load_golden('ddim_schedule_100')
    ts, a, ap, s1m = orc.ddim_schedule(ac, 100)
    assert (ts == g100['ddim_timesteps'].numpy()).all() and ts[0] == 1 and ts[-1] == 991
    assert torch.equal(a, g100['ddim_alphas']) and torch.equal(ap, g100['ddim_alphas_prev'].float())
    sd = _unet3d_sd(32, 64, 'unet3d_tiny.')
    z = orc.shape_sample_loop(sd, g['uc_s'], g['triples'], synth.shape_noise(seed=7), S=4)
    _close(z, g['z_final'], 2e-4)


@pytest.mark.slow
def test_unet3d_full():
    g = load_golden('unet3d_full')
    sd = _unet3d_sd(224, 1280, 'unet3d_full.')
    eps = orc.unet3d_forward(sd, g['x'], g['uc_s'], g['triples'], g['t'])
    _close(eps, g['eps'], 2e-4)


@pytest.mark.parametrize('tag', ['tiny', pytest.param('full', marks=pytest.mark.slow)])
def test_vqvae_decode(tag):
    from echoscene_amd.model.vqvae import VQVAE
    g = load_golden('vqvae_' + tag)
    ch, ne = [int(v) for v in g['cfg']]
    c = escfg.vqvae_conf(ch).model.params
    sd = seeded_state_dict(VQVAE(dict(c.ddconfig), ne, c.embed_dim), 'vqvae_%s.' % tag)
    _, idx = orc.vq_quantize(sd, g['z'])
    assert torch.equal(idx, g['idx'])
    sdf = orc.vqvae_decode_no_quant(sd, g['z'])
    assert tuple(sdf.shape[2:]) == (64, 64, 64)
    _close(sdf[:, :, ::4, ::4, ::4], g['sdf_sub'], 1e-4)
    assert abs(sdf.double().abs().sum().item() - g['sdf_abs'].item()) < 1e-4 * g['sdf_abs'].item()


@pytest.mark.parametrize('concat,gold', [(False, 'scene_e2e_tiny'), (True, 'scene_e2e_concat_tiny'), (False, 'scene_e2e_O2_tiny')])
def test_scene_e2e_tiny_oracle_vs_reference_api(concat, gold):
    """Whole boundary on the CPU oracle vs the reference's own ``SGDiff.sample_box_and_shape`` (tiny widths):
    setup GCNs -> 100-step layout loop -> rel_s_mlp -> 4-step DDIM -> VQ-VAE decode.  ``concat``: the
    config/full_concat_mp.yaml family (c_s enters the shape denoiser as an input channel).  ``scene_e2e_O2_tiny``: the
    smallest scene (one object + the scene node, one triple)."""
    from echoscene_amd.model.scene import SGDiff
    g = load_golden(gold)
    objs, triples = g['objs'], g['triples']
    O = objs.shape[0]
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
    for typ in (('echoscene',) if concat else ('echoscene', 'echolayout')):
        m = SGDiff(typ, escfg.tiny_diff_opt('cpu', concat=concat), synth.VOCAB, residual=True, with_angles=True)
        sd = seeded_state_dict(m.diff, 'e2e.diff.')
        oe, latent_m, _ = orc.scene_setup(sd, objs, triples, tf, rf, model_type=typ)
        lsd = {k[len('LayoutDiff.df.model.'):]: v for k, v in sd.items() if k.startswith('LayoutDiff.df.model.')}
        x = orc.layout_sample_loop(lsd, oe, triples, synth.layout_noise(O, 8, 100, seed=7), time_num=100)
        _close(x[:, 0:3], g[typ + '_sizes'], 2e-4)
        _close(x[:, 3:6], g[typ + '_translations'], 2e-4)
        _close(x[:, 6:8], g[typ + '_angles'], 2e-4)
        if typ == 'echoscene':
            uc = orc.rel_s(sd, oe)
            dsd = seeded_state_dict(m.diff.ShapeDiff.df, 'e2e.shape_df.')
            dsd = {k[len('diffusion_net.'):]: v for k, v in dsd.items()}
            z = orc.shape_sample_loop(dsd, uc, triples, synth.shape_noise(seed=7), S=4,
                                      c_concat=orc.rel_s(sd, latent_m) if concat else None)
            vsd = seeded_state_dict(m.diff.ShapeDiff.vqvae, 'e2e.vqvae.')
            sdf = orc.vqvae_decode_no_quant(vsd, z)
            assert tuple(sdf.shape) == (O, 1, 64, 64, 64)
            _close(sdf[:, :, ::4, ::4, ::4], g['echoscene_shapes'], 2e-3)


def _flags_opt(device):
    """tiny_diff_opt as tests/golden/make_golden.py case_scene_flags builds it: no ``enable_t_emb`` key (config/full.yaml), using_clip
    False (scripts/eval_3dfront.py:386)"""
    opt = escfg.tiny_diff_opt(device)
    del opt.layout_branch.denoiser_kwargs['enable_t_emb']
    opt.layout_branch.denoiser_kwargs.using_clip = False
    return opt


@pytest.mark.parametrize('typ', ['echoscene', 'echolayout'])
def test_scene_flag_matrix_oracle_vs_reference_api(typ):
    """The other corner of the SGDiff flag matrix (SGDiff.py:8-30) -- clip=False, residual=False, replace_latent=True, layout denoiser
    without enable_t_emb -- against the reference's own sample_box_and_shape and sample_boxes_and_shape_with_changes."""
    import numpy as np
    from echoscene_amd.model.scene import SGDiff
    g = load_golden('scene_flags_tiny')
    objs, triples = g['objs'], g['triples']
    O = objs.shape[0]
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
    m = SGDiff(typ, _flags_opt('cpu'), synth.VOCAB, replace_latent=True, residual=False, with_angles=True, clip=False)
    sd = seeded_state_dict(m.diff, 'e2e.diff.')
    assert not any('linear_projection' in k for k in sd if k.startswith('gconv_net_')), 'residual=False: no skip projections in the setup GCNs'
    lsd = {k[len('LayoutDiff.df.model.'):]: v for k, v in sd.items() if k.startswith('LayoutDiff.df.model.')}
    noise = synth.layout_noise(O, 8, 100, seed=7)
    manipulated = [int(v) for v in g['manipulated']]
    np.random.seed(123)
    change = torch.zeros(O, 64)
    for i in sorted(manipulated):
        change[i] = torch.from_numpy(np.random.normal(0, 1, 64)).float()
    for tag, ch in (('', None), ('chg_', change)):
        oe, latent_m, _ = orc.scene_setup(sd, objs, triples, tf, rf, model_type=typ, change_noise=ch, clip=False)
        assert oe.shape[1] == 128
        if ch is not None:                                   # replace_latent=True: every latent is the manipulator's (EchoScene.py:440-448)
            _close(latent_m, g[typ + '_chg_rel'], 2e-5)
        x = orc.layout_sample_loop(lsd, oe, triples, noise, time_num=100, enable_t_emb=False)
        _close(x[:, 0:3], g[typ + '_' + tag + 'sizes'], 2e-4)
        _close(x[:, 3:6], g[typ + '_' + tag + 'translations'], 2e-4)
        _close(x[:, 6:8], g[typ + '_' + tag + 'angles'], 2e-4)
        if typ == 'echoscene':
            dsd = seeded_state_dict(m.diff.ShapeDiff.df, 'e2e.shape_df.')
            dsd = {k[len('diffusion_net.'):]: v for k, v in dsd.items()}
            z = orc.shape_sample_loop(dsd, orc.rel_s(sd, oe), triples, synth.shape_noise(seed=7), S=4)
            _close(z, g['echoscene_' + tag + 'z'], 1e-3)                 # the latents the reference's DDIM loop handed to its VQ-VAE
            sdf = orc.vqvae_decode_no_quant(seeded_state_dict(m.diff.ShapeDiff.vqvae, 'e2e.vqvae.'), z)
            _close(sdf[:, :, ::4, ::4, ::4], g['echoscene_' + tag + 'shapes'], 2e-3)
    if typ == 'echoscene':
        keep = g['echoscene_chg_keep']
        assert [int(v) for v in keep.flatten()] == [0 if i in manipulated else 1 for i in range(O)]


# --------------------------------------------------------------------------------------------------------
# the 'concat'-conditioned model family (config/full_concat_mp.yaml,
# sdfusion-txt2shape_concat_mp.yaml): AttentionBlock / QKVAttentionLegacy, GCN output as input channels,
# full 3-D down/up-sampling (dims=4).  Goldens: tests/golden/make_golden.py case_concat (reference modules).
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,mc,cd', [('tiny', 128, 128), pytest.param('full', 512, 1280, marks=pytest.mark.slow)])
def test_unet1d_concat(tag, mc, cd):
    g = load_golden('unet1d_concat_' + tag)
    kw = dict(escfg.layout_denoiser_kwargs(mc, concat=True))
    kw['concat_dim'] = kw['crossattn_dim'] = cd
    sd = seeded_state_dict(UNet1DModel(**kw), 'unet1d_concat_%s.' % tag)
    eps = orc.unet1d_forward(sd, g['box'], g['obj_embed'], g['triples'], g['t'])
    _close(eps, g['eps'], 2e-5)
    if tag == 'tiny':
        noise = synth.layout_noise(8, 8, 100, seed=9)
        x = orc.layout_sample_loop(sd, g['loop_obj_embed'], g['loop_triples'], noise, 100, n_steps=10)
        _close(x, g['loop_x10'], 5e-5)


@pytest.mark.parametrize('tag,mc', [('tiny', 32), pytest.param('full', 224, marks=pytest.mark.slow)])
def test_unet3d_concat(tag, mc):
    g = load_golden('unet3d_concat_' + tag)
    df = DiffusionUNet(escfg.shape_unet_params(mc, concat=True), conditioning_key='concat')
    sd = {k[len('diffusion_net.'):]: v for k, v in seeded_state_dict(df, 'unet3d_concat_%s.' % tag).items()}
    eps = orc.unet3d_forward(sd, g['x'], g['uc_s'], g['triples'], g['t'], c_concat=g['c_s'])
    _close(eps, g['eps'], 5e-5)
    if tag == 'tiny':
        z = orc.shape_sample_loop(sd, g['uc_s'], g['triples'], synth.shape_noise(seed=7), S=4, c_concat=g['c_s'])
        _close(z, g['z_final'], 2e-4)


@pytest.mark.parametrize('fam', ['crossattn', 'concat'])
def test_unet3d_without_message_passing(fam):
    """config/sdfusion-txt2shape.yaml / sdfusion-txt2shape_concat.yaml: no GCN, c_s is the key / the concat channel."""
    g = load_golden('unet3d_nomp_' + fam)
    p = escfg.shape_unet_params(32, concat=(fam == 'concat'), mp=False)
    if fam == 'crossattn':
        p['context_dim'] = 64
    df = DiffusionUNet(p, conditioning_key=fam)
    sd = {k[len('diffusion_net.'):]: v for k, v in seeded_state_dict(df, 'unet3d_nomp_%s.' % fam).items()}
    kw = dict(c_concat=g['c_s']) if fam == 'concat' else dict(context=g['c_s'])
    eps = orc.unet3d_forward(sd, g['x'], g['uc_s'], g['triples'], g['t'], **kw)
    _close(eps, g['eps'], 5e-5)
    z = orc.shape_sample_loop(sd, g['uc_s'], g['triples'], synth.shape_noise(seed=7), S=4, **kw)
    _close(z, g['z_final'], 2e-4)


def test_box_postprocess_helpers_vs_reference():
    """descale_box_params / postprocess_sincos2arctan restated in the oracle vs the reference's own helpers."""
    g = load_golden('box_post')
    out = orc.descale_box_params(g['boxes'].clone(), g['stats'].numpy())
    _close(torch.as_tensor(out), g['boxes_out'], 1e-6)
    out7 = orc.descale_box_params(g['boxes7'].clone(), g['stats'].numpy(), angle=True)         # helpers/util.py:553-555
    _close(torch.as_tensor(out7), g['boxes7_out'], 1e-6)
    _close(torch.as_tensor(orc.sincos2arctan(g['sincos'])).reshape(-1, 1), g['angle'], 1e-6)
