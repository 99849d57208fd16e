#!/usr/bin/env python
"""Golden-vector generator: runs the REFERENCE ITSELF (read-only, from /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py [--only NAME] [--ref /root/reference]

For every case it (1) instantiates the reference's own module, (2) overwrites all of its
parameters/buffers with ``echoscene_amd.synth.seeded_tensor`` (so the consumer can
regenerate identical weights from names+shapes; zero-initialised tensors are re-randomised,
BatchNorm runs on running statistics), (3) feeds seeded inputs, (4) stores inputs-by-value
(small) and outputs in ``tests/golden/<case>.npz``.  Nothing from the reference's source is
copied; the .npz files contain numbers only.

Packages the reference imports for rendering / datasets / training that are absent from the
image (trimesh, pytorch3d, cv2, mcubes, termcolor, torchvision, fvcore, omegaconf, ...) are
replaced by inert stand-ins -- none of them is on the numeric path.
"""
import argparse
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from echoscene_amd import synth, config as escfg  # noqa: E402

MISSING = ('trimesh', 'pytorch3d', 'cv2', 'mcubes', 'termcolor', 'torchvision', 'fvcore', 'open3d',
           'h5py', 'imageio', 'skimage', 'tensorboardX', 'clip', 'pyrender', 'seaborn', 'omegaconf')


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        cls = type(name, (object,), {'__init__': lambda self, *a, **k: None,
                                     '__call__': lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        name = module.__name__
        if name == 'omegaconf':
            module.OmegaConf = type('OmegaConf', (), {
                'load': staticmethod(lambda p: escfg.resolve_nested(p)),
                'create': staticmethod(lambda d: escfg.to_plain(d))})
        if name == 'omegaconf.listconfig':
            module.ListConfig = type('ListConfig', (list,), {})
        if name == 'termcolor':
            module.colored = lambda s, *a, **k: s
            module.cprint = lambda s, *a, **k: print(s)


def install_reference(ref):
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, ref)
    # the reference hard-codes .cuda() / device='cuda' in a few places (ddim.py:22-26,
    # echo2shape.py:509); this container has no GPU.
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _randn = torch.randn

    def randn(*a, **k):
        if str(k.get('device', '')).startswith('cuda'):
            k['device'] = 'cpu'
        return _randn(*a, **k)
    torch.randn = randn


def fill(module, prefix, seed=0):
    synth.seeded_fill_(module, seed=seed, prefix=prefix)
    module.eval()
    return module


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------
def case_gcn():
    from model.graph import GraphTripleConvNet
    objs, triples = synth.synthetic_graph(8, seed=1)
    for tag, residual, norm in (('res_bn', True, 'batch'), ('plain', False, 'none')):
        net = GraphTripleConvNet(input_dim_obj=96, input_dim_pred=32, num_layers=3, hidden_dim=64,
                                 residual=residual, pooling='avg', mlp_normalization=norm, output_dim=80)
        fill(net, 'gcn_%s.' % tag)
        obj = rnd((8, 96), 11)
        pred = rnd((triples.shape[0], 32), 12)
        edges = torch.stack([triples[:, 0], triples[:, 2]], 1)
        with torch.no_grad():
            o, p = net(obj, pred, edges)
        save('gcn_' + tag, obj=obj, pred=pred, triples=triples, out_obj=o, out_pred=p,
             cfg=np.array([96, 32, 3, 64, int(residual), int(norm == 'batch'), 80]))


def _unet1d(model_channels, ctx_dim, time_num=1000, concat=False):
    from model.networks.diffusion_layout.denoise_net import UNet1DModel
    kw = dict(escfg.layout_denoiser_kwargs(model_channels, concat=concat))
    kw['concat_dim'] = kw['crossattn_dim'] = ctx_dim
    return UNet1DModel(**kw), kw


def case_unet1d_tiny():
    net, kw = _unet1d(128, 128)
    fill(net, 'unet1d_tiny.')
    objs, triples = synth.synthetic_graph(8, seed=2)
    box = rnd((8, 8), 21)
    oe = rnd((8, 640), 22)
    t = torch.full((8,), 437, dtype=torch.int64)
    with torch.no_grad():
        eps = net(box, oe, triples, t)
    save('unet1d_tiny', box=box, obj_embed=oe, triples=triples, t=t, eps=eps.squeeze(-1))


def _layout_loop(net, kw, O, seed_graph, time_num, n_steps, noise, force_traj=False, clip_denoised=False, **diffusion_overrides):
    """Runs the reference's own DiffusionPoint / GaussianDiffusion.p_sample_loop_sg with an
    injected noise_fn; optionally truncated to the first n_steps iterations."""
    from model.networks.diffusion_layout.diffusion_ddpm import DiffusionPoint
    cfg = escfg.AttrDict(angle_dim=2)
    dkw = dict(escfg.layout_diffusion_kwargs(time_num))
    dkw.update(diffusion_overrides)
    df = DiffusionPoint(denoise_net=net, config=cfg, **dkw)
    objs, triples = synth.synthetic_graph(O, seed=seed_graph)
    oe = rnd((O, 640), 100 + seed_graph)
    calls = {'n': 0}

    def noise_fn(size, dtype, device):
        i = calls['n']
        calls['n'] += 1
        return noise[i].clone()

    gd = df.diffusion
    traj = []
    with torch.no_grad():
        if n_steps == time_num and not force_traj:
            x = df.gen_samples_sg((O, 8), 'cpu', oe, triples, condition=None, noise_fn=noise_fn,
                                  clip_denoised=clip_denoised)
        else:
            # same body as p_sample_loop_sg, stopped early: call the reference's p_sample_sg
            x = noise_fn(size=(O, 8), dtype=torch.float, device='cpu')
            for t in list(reversed(range(time_num)))[:n_steps]:
                t_ = torch.empty(O, dtype=torch.int64).fill_(t)
                x = gd.p_sample_sg(denoise_fn=df._denoise, data=x, t=t_, obj_embed=oe, triples=triples,
                                   condition=None, noise_fn=noise_fn, clip_denoised=clip_denoised)
                traj.append(x.clone())
    tabs = {k: getattr(gd, k) for k in ('sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod',
                                        'posterior_mean_coef1', 'posterior_mean_coef2',
                                        'posterior_log_variance_clipped')}
    return oe, triples, x, traj, tabs


def case_layout_loop_tiny():
    """BASELINE.json configs[0]: box-only diffusion, 8-node graph, 100 DDPM steps (tiny width)."""
    net, kw = _unet1d(128, 128)
    fill(net, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    oe, triples, x, _, tabs = _layout_loop(net, kw, 8, 3, 100, 100, noise)
    save('layout_loop_tiny', obj_embed=oe, triples=triples, x_final=x,
         **{'tab100_' + k: v for k, v in tabs.items()})


def case_ddpm_tables():
    from model.networks.diffusion_layout.diffusion_ddpm import GaussianDiffusion, get_betas
    gd = GaussianDiffusion(escfg.AttrDict(), get_betas('linear', 1e-4, 0.02, 1000), 'mse', 'eps',
                           'fixedsmall', True, False, 'obb', None)
    save('ddpm_tables_1000', **{k: getattr(gd, k) for k in (
        'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_mean_coef1',
        'posterior_mean_coef2', 'posterior_log_variance_clipped')})


def case_unet1d_full():
    """Full-width layout denoiser (config/full_mp.yaml): one forward at O=8 and O=32 and a
    10-step trajectory of the 1000-step loop at O=8.  Weights regenerate from the seeded rule."""
    net, kw = _unet1d(512, 1280)
    fill(net, 'unet1d_full.')
    out = {}
    for O, sg in ((8, 4), (32, 5)):
        objs, triples = synth.synthetic_graph(O, seed=sg)
        box = rnd((O, 8), 30 + O)
        oe = rnd((O, 640), 40 + O)
        t = torch.full((O,), 617, dtype=torch.int64)
        with torch.no_grad():
            eps = net(box, oe, triples, t).squeeze(-1)
        out.update({'box%d' % O: box, 'obj_embed%d' % O: oe, 'triples%d' % O: triples, 'eps%d' % O: eps})
    noise = synth.layout_noise(8, 8, 1000, seed=7)[:11]
    oe, triples, x, traj, _ = _layout_loop(net, kw, 8, 4, 1000, 10, noise)
    out.update({'loop_obj_embed': oe, 'loop_triples': triples, 'loop_x10': x, 'loop_x1': traj[0]})
    save('unet1d_full', **out)


def _unet3d(model_channels, ctx_dim, concat=False):
    from model.networks.diffusion_shape.network import DiffusionUNet
    p = escfg.shape_unet_params(model_channels, concat=concat)
    if not concat:
        p['context_dim'] = ctx_dim
    return DiffusionUNet(p, vq_conf=None, conditioning_key='concat' if concat else 'crossattn')


def case_unet3d_tiny():
    net = _unet3d(32, 64)
    fill(net, 'unet3d_tiny.')
    O = 4
    objs, triples = synth.synthetic_graph(O, seed=6)
    x = rnd((O, 3, 16, 16, 16), 51)
    uc = rnd((O, 1, 64), 52)
    t = torch.full((O,), 401, dtype=torch.long)
    with torch.no_grad():
        eps = net(x, uc, triples, t, c_crossattn=[rnd((O, 1, 64), 53)])
    save('unet3d_tiny', x=x, uc_s=uc, triples=triples, t=t, eps=eps)


class _ShapeShim:
    """Carries the attributes EchoToShape.register_schedule / apply_model / DDIMSampler touch, so
    the reference's *own* schedule, apply_model and DDIM code run without constructing the whole
    EchoToShape (which needs a VQ-VAE checkpoint file and a mesh renderer)."""
    parameterization = 'eps'
    v_posterior = 0.
    device = 'cpu'


def case_ddim_tiny():
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    net = _unet3d(32, 64)
    fill(net, 'unet3d_tiny.')
    shim = _ShapeShim()
    shim.df = shim.df_module = net
    EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
    shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)   # cuda-free
    O = 4
    objs, triples = synth.synthetic_graph(O, seed=6)
    uc = rnd((O, 1, 64), 52)
    c = rnd((O, 1, 64), 53)
    noise1 = synth.shape_noise(seed=7)
    sampler = DDIMSampler(shim)
    with torch.no_grad():
        z, inter = sampler.sample(S=4, batch_size=O, shape=(3, 16, 16, 16), conditioning=c,
                                  x_T=noise1.repeat(O, 1, 1, 1, 1), verbose=False,
                                  unconditional_guidance_scale=3., unconditional_conditioning=uc,
                                  triplet=triples, eta=0.0)
    save('ddim_tiny', uc_s=uc, triples=triples, z_final=z, alphas_cumprod=shim.alphas_cumprod,
         ddim_timesteps=sampler.ddim_timesteps, ddim_alphas=sampler.ddim_alphas,
         ddim_alphas_prev=sampler.ddim_alphas_prev,
         ddim_sqrt_one_minus_alphas=sampler.ddim_sqrt_one_minus_alphas)
    # the shipped S=100 schedule tables (reference behaviour)
    s100 = DDIMSampler(shim)
    s100.make_schedule(ddim_num_steps=100, ddim_eta=0.0, verbose=False)
    save('ddim_schedule_100', ddim_timesteps=s100.ddim_timesteps, ddim_alphas=s100.ddim_alphas,
         ddim_alphas_prev=s100.ddim_alphas_prev,
         ddim_sqrt_one_minus_alphas=s100.ddim_sqrt_one_minus_alphas)


def case_unet3d_full():
    net = _unet3d(224, 1280)
    fill(net, 'unet3d_full.')
    O = 2
    objs, triples = synth.synthetic_graph(O, seed=8)
    x = rnd((O, 3, 16, 16, 16), 61)
    uc = rnd((O, 1, 1280), 62)
    t = torch.full((O,), 401, dtype=torch.long)
    with torch.no_grad():
        eps = net(x, uc, triples, t, c_crossattn=[uc])
    save('unet3d_full', x=x, uc_s=uc, triples=triples, t=t, eps=eps)


def case_concat():
    """the 'concat'-conditioned model family (config/full_concat_mp.yaml,
    sdfusion-txt2shape_concat_mp.yaml) -- reference modules, seeded weights, tiny and full widths."""
    # layout denoiser: eps + 10 ancestral steps
    for tag, mc, cd, O in (('tiny', 128, 128, 8), ('full', 512, 1280, 6)):
        net, kw = _unet1d(mc, cd, concat=True)
        fill(net, 'unet1d_concat_%s.' % tag)
        objs, triples = synth.synthetic_graph(O, seed=12)
        box = rnd((O, 8), 121)
        oe = rnd((O, 640), 122)
        t = torch.full((O,), 437, dtype=torch.int64)
        with torch.no_grad():
            eps = net(box, oe, triples, t)
        out = dict(box=box, obj_embed=oe, triples=triples, t=t, eps=eps.squeeze(-1))
        if tag == 'tiny':
            noise = synth.layout_noise(8, 8, 100, seed=9)
            oe2, tri2, x, traj, _ = _layout_loop(net, kw, 8, 13, 100, 10, noise)
            out.update(loop_obj_embed=oe2, loop_triples=tri2, loop_x10=x)
        save('unet1d_concat_' + tag, **out)
    # shape denoiser: eps (tiny + full) and a 4-step DDIM loop (tiny) through the reference's own sampler
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    for tag, mc, O in (('tiny', 32, 3), ('full', 224, 2)):
        net = _unet3d(mc, None, concat=True)
        fill(net, 'unet3d_concat_%s.' % tag)
        objs, triples = synth.synthetic_graph(O, seed=14)
        x = rnd((O, 3, 16, 16, 16), 151)
        uc = rnd((O, 1, 4096), 152)
        c = rnd((O, 1, 16, 16, 16), 153)
        t = torch.full((O,), 401, dtype=torch.long)
        with torch.no_grad():
            eps = net(x, uc, triples, t, c_concat=[c])
        out = dict(x=x, uc_s=uc, c_s=c, triples=triples, t=t, eps=eps)
        if tag == 'tiny':
            shim = _ShapeShim()
            shim.df = shim.df_module = net
            EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
            shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
            DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
            noise1 = synth.shape_noise(seed=7)
            with torch.no_grad():
                z, _ = DDIMSampler(shim).sample(S=4, batch_size=O, shape=(3, 16, 16, 16), conditioning=c,
                                                x_T=noise1.repeat(O, 1, 1, 1, 1), verbose=False,
                                                unconditional_guidance_scale=3., unconditional_conditioning=uc,
                                                triplet=triples, eta=0.0)
            out.update(z_final=z)
        save('unet3d_concat_' + tag, **out)


def case_layout_traj_full():
    """VERDICT r1 #1 / SURVEY 8(c): the benchmarked layout configuration (BASELINE configs[1]): full width, O=32, the
    reference's own p_sample_sg for all 1000 ancestral steps; the whole trajectory is stored ([1000,32,8] f32 = 1 MB)."""
    net, kw = _unet1d(512, 1280)
    fill(net, 'unet1d_full.')
    noise = synth.layout_noise(32, 8, 1000, seed=7)
    oe, triples, x, traj, _ = _layout_loop(net, kw, 32, 5, 1000, 1000, noise, force_traj=True)
    save('layout_traj_full', obj_embed=oe, triples=triples, traj=torch.stack(traj), x_final=x)


def case_shape_traj_full():
    """VERDICT r1 #1 / SURVEY 8(c): the benchmarked shape configuration: model_channels 224, the reference's own
    DDIMSampler for all 100 steps (eta 0) at O=4, then the reference VQ-VAE decode_no_quant at full size (8192 codes).
    Stored: z after 1/2/5/10/20/50/100 steps, per-step mean|z| and rms for all steps, the VQ indices, the occupancy
    sdf<0.02 as a packed bit mask (full 64^3 resolution) and a ::2 sub-sample of the SDF."""
    import time
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    net = _unet3d(224, 1280)
    fill(net, 'unet3d_full.')
    shim = _ShapeShim()
    shim.df = shim.df_module = net
    EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
    shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    O = 4
    objs, triples = synth.synthetic_graph(O, seed=18)
    uc = rnd((O, 1, 1280), 182)
    noise1 = synth.shape_noise(seed=7)
    t0 = time.time()
    with torch.no_grad():
        z, inter = DDIMSampler(shim).sample(S=100, batch_size=O, shape=(3, 16, 16, 16), conditioning=uc,
                                            x_T=noise1.repeat(O, 1, 1, 1, 1), verbose=False, log_every_t=1,
                                            unconditional_guidance_scale=3., unconditional_conditioning=uc,
                                            triplet=triples, eta=0.0)
    print('100 DDIM steps: %.0f s' % (time.time() - t0))
    xs = inter['x_inter']            # [x_T, after step 1, ..., after step 100]
    assert len(xs) == 101 and torch.equal(xs[-1], z)
    keep = (1, 2, 5, 10, 20, 50, 100)
    vq = _vqvae(64, 8192)
    fill(vq, 'vqvae_full.')
    with torch.no_grad():
        _, _, (_, _, idx) = vq.quantize(z, is_voxel=True)
        sdf = vq.decode_no_quant(z)
    occ = np.packbits((sdf.numpy() < 0.02).reshape(-1))
    save('shape_traj_full', uc_s=uc, triples=triples, steps=np.array(keep),
         z_steps=torch.stack([xs[k] for k in keep]),
         z_mean_abs=np.array([float(x.abs().mean()) for x in xs]),
         z_rms=np.array([float(x.pow(2).mean().sqrt()) for x in xs]),
         vq_idx=idx.reshape(-1), occ_bits=occ, sdf_sub=sdf[:, :, ::2, ::2, ::2],
         sdf_sum=sdf.double().sum(), sdf_abs=sdf.double().abs().sum())


def case_unet3d_full_O32():
    """ONE reference UNet3D + echo-GCN eps evaluation at the BENCHMARKED shape (model_channels 224,
    O = 32: every 3x3x3 / 1x1 launch has >= 256 row tiles, i.e. the plain k_conv_ws / k_linear_ws routes the bench times).
    Same weights ('unet3d_full.') as the O = 2 / O = 4 goldens; openai_model_3d.py:816-863."""
    import time
    net = _unet3d(224, 1280)
    fill(net, 'unet3d_full.')
    O = 32
    objs, triples = synth.synthetic_graph(O, seed=5)
    x = rnd((O, 3, 16, 16, 16), 611)
    uc = rnd((O, 1, 1280), 612)
    t = torch.full((O,), 401, dtype=torch.long)
    t0 = time.time()
    with torch.no_grad():
        eps = net(x, uc, triples, t, c_crossattn=[uc])
    print('UNet3D forward at O=32: %.0f s' % (time.time() - t0))
    save('unet3d_full_O32', x_seed=np.array(611), uc_s=uc, triples=triples, t=t, eps=eps)   # x = rnd((O,3,16,16,16), x_seed)


def case_shape_traj_full_O16():
    """VERDICT r2 #1 / BASELINE configs[2]: 2 DDIM steps of the reference's own DDIMSampler at model_channels 224, O = 16
    (16x8x8 level = 128 row tiles -> split-K dispatch differs from both O = 4 and O = 32); ddim.py:127-262."""
    import time
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    net = _unet3d(224, 1280)
    fill(net, 'unet3d_full.')
    shim = _ShapeShim()
    shim.df = shim.df_module = net
    EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
    shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    O = 16
    objs, triples = synth.synthetic_graph(O, seed=16)
    uc = rnd((O, 1, 1280), 162)
    noise1 = synth.shape_noise(seed=7)
    sampler = DDIMSampler(shim)
    sampler.make_schedule(ddim_num_steps=100, ddim_eta=0.0, verbose=False)
    x = noise1.repeat(O, 1, 1, 1, 1)
    ts = np.flip(sampler.ddim_timesteps)
    zs = []
    t0 = time.time()
    with torch.no_grad():
        for i in range(2):
            index = len(ts) - i - 1
            tt = torch.full((O,), int(ts[i]), dtype=torch.long)
            x, _ = sampler.p_sample_ddim(x, uc, tt, index=index, unconditional_guidance_scale=3.,
                                         unconditional_conditioning=uc, triplet=triples)
            zs.append(x.clone())
    print('2 DDIM steps at O=16: %.0f s' % (time.time() - t0))
    save('shape_traj_full_O16', uc_s=uc, triples=triples, z_steps=torch.stack(zs))


def _vqvae(ch, n_embed):
    from model.networks.vqvae_networks.network import VQVAE
    p = escfg.vqvae_conf(ch).model.params
    p.n_embed = n_embed
    return VQVAE(dict(p.ddconfig), p.n_embed, p.embed_dim)


def case_vqvae():
    for tag, ch, ne, B in (('tiny', 32, 64, 2), ('full', 64, 8192, 1)):
        vq = _vqvae(ch, ne)
        fill(vq, 'vqvae_%s.' % tag)
        z = rnd((B, 3, 16, 16, 16), 71, 0.6)
        with torch.no_grad():
            _, _, (_, _, idx) = vq.quantize(z, is_voxel=True)
            sdf = vq.decode_no_quant(z)
        save('vqvae_' + tag, z=z, idx=idx.reshape(-1), sdf_sub=sdf[:, :, ::4, ::4, ::4],
             sdf_sum=sdf.double().sum(), sdf_abs=sdf.double().abs().sum(), cfg=np.array([ch, ne]))


def case_box_post():
    """the reference's own helpers/util.py descale_box_params / postprocess_sincos2arctan on random boxes."""
    import tempfile
    from helpers.util import descale_box_params, postprocess_sincos2arctan
    rs = np.random.RandomState(0)
    boxes = torch.from_numpy(rs.uniform(-1.2, 1.2, (33, 6)).astype(np.float32))
    sc = torch.from_numpy(rs.standard_normal((33, 2)).astype(np.float32))
    stats = np.concatenate([rs.uniform(0.1, 0.5, 3), rs.uniform(1.0, 3.0, 3), rs.uniform(-4, -2, 3), rs.uniform(2, 4, 3),
                            [-np.pi, np.pi]])
    f = os.path.join(tempfile.mkdtemp(prefix='golden_box_'), 'stats.txt')
    np.savetxt(f, stats)
    out = descale_box_params(boxes.clone(), file=f)
    ang = postprocess_sincos2arctan(sc)
    boxes7 = torch.from_numpy(rs.uniform(-1.2, 1.2, (33, 7)).astype(np.float32))         # angle=True: 7-column boxes
    out7 = descale_box_params(boxes7.clone(), file=f, angle=True)
    save('box_post', boxes=boxes, sincos=sc, stats=torch.from_numpy(stats), boxes_out=out, angle=ang, boxes7=boxes7, boxes7_out=out7)


def case_nomp():
    """Shape denoisers WITHOUT echo message passing (config/sdfusion-txt2shape.yaml, sdfusion-txt2shape_concat.yaml):
    objects independent, c_s is the cross-attention key / the concat channel.  eps + 4-step DDIM loop, tiny widths."""
    from model.networks.diffusion_shape.network import DiffusionUNet
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    for fam in ('crossattn', 'concat'):
        p = escfg.shape_unet_params(32, concat=(fam == 'concat'), mp=False)
        if fam == 'crossattn':
            p['context_dim'] = 64
        net = DiffusionUNet(p, vq_conf=None, conditioning_key=fam)
        fill(net, 'unet3d_nomp_%s.' % fam)
        O = 3
        objs, triples = synth.synthetic_graph(O, seed=16)
        x = rnd((O, 3, 16, 16, 16), 171)
        uc = rnd((O, 1, 64), 172)
        c = rnd((O, 1, 64), 173) if fam == 'crossattn' else rnd((O, 1, 16, 16, 16), 173)
        t = torch.full((O,), 401, dtype=torch.long)
        with torch.no_grad():
            eps = net(x, uc, triples, t, **({'c_crossattn': [c]} if fam == 'crossattn' else {'c_concat': [c]}))
        shim = _ShapeShim()
        shim.df = shim.df_module = net
        EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
        shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
        DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
        noise1 = synth.shape_noise(seed=7)
        with torch.no_grad():
            z, _ = DDIMSampler(shim).sample(S=4, batch_size=O, shape=(3, 16, 16, 16), conditioning=c,
                                            x_T=noise1.repeat(O, 1, 1, 1, 1), verbose=False,
                                            unconditional_guidance_scale=3., unconditional_conditioning=uc,
                                            triplet=triples, eta=0.0)
        save('unet3d_nomp_' + fam, x=x, uc_s=uc, c_s=c, triples=triples, t=t, eps=eps, z_final=z)


class _SGDiffHarness:
    """The reference's own SGDiff on CPU (SURVEY.md 8(c) recipe) with both loops' noise injected; shared by the
    end-to-end and the editing cases."""

    def __init__(self, typ, concat, O=8, graph_seed=9, clip=True, residual=True, replace_latent=False, t_emb=True):
        """``clip`` / ``residual`` / ``replace_latent``: the SGDiff constructor flags (SGDiff.py:8-9); ``clip=False`` also sets
        ``denoiser_kwargs.using_clip`` as scripts/eval_3dfront.py:386 does.  ``t_emb=False``: the layout denoiser_kwargs carry no
        ``enable_t_emb`` key (config/box.yaml, config/full.yaml): UNet1DModel then builds without box_time_emb (denoise_net.py:505)."""
        import tempfile
        self.tmp = tempfile.mkdtemp(prefix='golden_e2e_')
        vq = _vqvae(32, 64)
        fill(vq, 'e2e.vqvae.')
        vq_path = os.path.join(self.tmp, 'vq.pth')
        torch.save(vq.state_dict(), vq_path)
        opt = escfg.tiny_diff_opt(device='cpu', logs_dir=self.tmp, vq_ckpt=vq_path, concat=concat)
        opt.misc.debug = 0
        if not t_emb:
            del opt.layout_branch.denoiser_kwargs['enable_t_emb']
        opt.layout_branch.denoiser_kwargs.using_clip = clip              # scripts/eval_3dfront.py:386
        import model.networks.diffusion_shape.echo2shape as e2s
        e2s.init_mesh_renderer = lambda **k: None
        from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
        DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
        from model.SGDiff import SGDiff
        m = SGDiff(typ, opt, synth.VOCAB, replace_latent=replace_latent, with_changes=True, residual=residual,
                   gconv_pooling='avg', with_angles=True, clip=clip, separated=False)
        synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), seed=0, prefix='e2e.diff.')
        if typ == 'echoscene':
            fill(m.diff.ShapeDiff.df, 'e2e.shape_df.')
            m.diff.ShapeDiff.ddim_steps = 4
        m.eval()
        self.m, self.typ, self.O = m, typ, O
        self.objs, self.triples = synth.synthetic_graph(O, seed=graph_seed)
        self.tf, self.rf = synth.synthetic_features(O, self.triples.shape[0], seed=graph_seed)
        self.noise = synth.layout_noise(O, 8, 100, seed=7)
        self.noise1 = synth.shape_noise(seed=7)

    def call(self, fn):
        """run fn() with the layout loop's noise_fn and rel2shape's torch.randn replaced by the seeded tensors"""
        import model.networks.diffusion_layout.diffusion_ddpm as dd
        calls = {'n': 0}
        noise, noise1 = self.noise, self.noise1
        _orig_gen = dd.DiffusionPoint.gen_samples_sg

        def noise_fn(size, dtype, device):
            i = calls['n']
            calls['n'] += 1
            return noise[i].clone()

        def gen(self_, shape, device, obj_embed, triples=None, condition=None, noise_fn_=None, clip_denoised=True,
                keep_running=False, **kw):
            return _orig_gen(self_, shape, device, obj_embed, triples, condition=condition, noise_fn=noise_fn,
                             clip_denoised=clip_denoised, keep_running=keep_running)
        dd.DiffusionPoint.gen_samples_sg = gen
        _randn = torch.randn

        def randn(*a, **k):
            size = k.get('size', a[0] if len(a) == 1 and not isinstance(a[0], int) else a)
            if tuple(size) == (1, 3, 16, 16, 16):
                return noise1.clone()
            return _randn(*a, **k)
        torch.randn = randn
        # the latents the DDIM loop hands to the VQ-VAE (rel2shape -> vqvae.decode_no_quant, echo2shape.py:521-522): recorded so that a
        # consumer can be compared BEFORE the codebook argmin, which turns a 4th-digit difference into an O(1) local change of the SDF
        self.last_z = None
        vq = self.m.diff.ShapeDiff.vqvae if self.typ == 'echoscene' else None
        _dec = vq.decode_no_quant if vq is not None else None
        if vq is not None:
            def dec(z, *a, **k):
                self.last_z = z.detach().clone()
                return _dec(z, *a, **k)
            vq.decode_no_quant = dec
        try:
            with torch.no_grad():
                r = fn()
        finally:
            torch.randn = _randn
            dd.DiffusionPoint.gen_samples_sg = _orig_gen
            if vq is not None:
                vq.decode_no_quant = _dec
        assert calls['n'] == 101, calls
        return r


def case_scene_edit():
    """SURVEY 8(f1) / VERDICT r1 #2: the reference's own sample_boxes_and_shape_with_changes / _with_additions
    (EchoScene.py:422-532, EchoLayout.py:309-401) for echoscene ('crossattn' and 'concat' families, gen_shape=True) and
    echolayout, tiny widths, numpy RNG seeded right before each call (the 64-d change noise, EchoScene.py:428-435).
    Two manipulated / two missing nodes so that the row bookkeeping (missing[i]+i, nodes_added vs missing_nodes) shows.
    Also stored: the conditioning the manipulator produced (LayoutDiff.rel = relation_cond), because the shipped
    'crossattn'+mp denoisers overwrite their context and would hide a wrong manipulator input."""
    out = {}
    manipulated = [5, 2]                 # unsorted on purpose
    missing = [2, 4]                     # -> nodes_added = [2, 5]
    for fam, typ, concat in (('sc', 'echoscene', False), ('cat', 'echoscene', True), ('lay', 'echolayout', False)):
        h = _SGDiffHarness(typ, concat)
        m = h.m
        dec = (h.objs, h.triples, h.tf, h.rf)
        # changes: same graph on both sides, two nodes get change noise
        np.random.seed(123)
        r = h.call(lambda: m.sample_boxes_and_shape_with_changes(*dec, *dec, manipulated, **(
            {} if typ == 'echolayout' else {'gen_shape': True})))
        keep, d = r
        out['%s_chg_keep' % fam] = keep
        out['%s_chg_rel' % fam] = m.diff.LayoutDiff.rel
        for k, v in d.items():
            if v is not None:
                out['%s_chg_%s' % (fam, k)] = v[:, :, ::4, ::4, ::4] if k == 'shapes' else v
                if k == 'shapes':
                    out['%s_chg_shapes_abs' % fam] = v.double().abs().sum()
        # additions: the encoder sees the graph without the two missing nodes
        added = [mi + i for i, mi in enumerate(missing)]
        eo, et, keep_idx, keep_tri = synth.remove_nodes(h.objs, h.triples, added)
        enc = (eo, et, h.tf[keep_idx], h.rf[keep_tri])
        np.random.seed(321)
        r = h.call(lambda: m.sample_boxes_and_shape_with_additions(*enc, *dec, missing, **(
            {} if typ == 'echolayout' else {'gen_shape': True})))
        if typ == 'echolayout':
            d = r                       # sic: the facade drops ``keep`` (SGDiff.py:113-115)
        else:
            keep, d = r
            out['%s_add_keep' % fam] = keep
        out['%s_add_rel' % fam] = m.diff.LayoutDiff.rel
        for k, v in d.items():
            if v is not None:
                out['%s_add_%s' % (fam, k)] = v[:, :, ::4, ::4, ::4] if k == 'shapes' else v
                if k == 'shapes':
                    out['%s_add_shapes_abs' % fam] = v.double().abs().sum()
        out.update(objs=h.objs, triples=h.triples)
    out.update(manipulated=np.array(manipulated), missing=np.array(missing))
    save('scene_edit_tiny', **out)


def case_unet1d_no_temb():
    """the layout denoiser as config/box.yaml and config/full.yaml build it -- their denoiser_kwargs carry NO
    ``enable_t_emb`` key, so UNet1DModel takes its default False (denoise_net.py:505): no box_time_emb, the box GCN's node
    vectors are [obj_embed 640 | box 64] (denoise_net.py:735-740, 758-771).  Full width: one forward at O = 8 and O = 32 and 10
    steps of the 1000-step loop; tiny width: the whole 100-step loop (BASELINE configs[0] shape)."""
    from model.networks.diffusion_layout.denoise_net import UNet1DModel
    for tag, mc, ctx in (('full', 512, 1280), ('tiny', 128, 128)):
        kw = dict(escfg.layout_denoiser_kwargs(mc))
        kw['concat_dim'] = kw['crossattn_dim'] = ctx
        del kw['enable_t_emb']
        net = UNet1DModel(**kw)
        assert not net.enable_t_emb and not hasattr(net, 'box_time_emb')
        fill(net, 'unet1d_%s_no_temb.' % tag)
        out = {}
        if tag == 'full':
            for O, sg in ((8, 4), (32, 5)):
                objs, triples = synth.synthetic_graph(O, seed=sg)
                box = rnd((O, 8), 30 + O)
                oe = rnd((O, 640), 40 + O)
                t = torch.full((O,), 617, dtype=torch.int64)
                with torch.no_grad():
                    eps = net(box, oe, triples, t).squeeze(-1)
                out.update({'box%d' % O: box, 'obj_embed%d' % O: oe, 'triples%d' % O: triples, 'eps%d' % O: eps})
            noise = synth.layout_noise(8, 8, 1000, seed=7)[:11]
            oe, triples, x, traj, _ = _layout_loop(net, kw, 8, 4, 1000, 10, noise)
            out.update({'loop_obj_embed': oe, 'loop_triples': triples, 'loop_x10': x, 'loop_x1': traj[0]})
            save('unet1d_full_no_temb', **out)
        else:
            noise = synth.layout_noise(8, 8, 100, seed=7)
            oe, triples, x, _, _ = _layout_loop(net, kw, 8, 3, 100, 100, noise)
            save('layout_loop_tiny_no_temb', obj_embed=oe, triples=triples, x_final=x)


def case_scene_flags():
    """the other corner of the SGDiff flag matrix (SGDiff.py:8-30): ``clip=False`` (no CLIP features: node /
    predicate vectors are the 128-d embeddings only, EchoScene.py:45-73,151-153; denoiser_kwargs.using_clip False as
    eval_3dfront.py:386 sets it), ``residual=False`` (setup GCNs without the skip projections, model/graph.py:205-209),
    ``replace_latent=True`` (editing: ALL latents come from the manipulator, EchoScene.py:440-448) and a layout denoiser
    without ``enable_t_emb`` (config/full.yaml / box.yaml).  Through the reference's own API: sample_box_and_shape for both model
    types and sample_boxes_and_shape_with_changes for echoscene (gen_shape=True)."""
    out = {}
    manipulated = [5, 2]
    for typ in ('echoscene', 'echolayout'):
        h = _SGDiffHarness(typ, False, clip=False, residual=False, replace_latent=True, t_emb=False)
        m = h.m
        assert not m.diff.clip and m.diff.replace_all_latent
        d = h.call(lambda: m.sample_box_and_shape(h.objs, h.triples, h.tf, h.rf, gen_shape=(typ == 'echoscene')))
        for k, v in d.items():
            if v is not None:
                out['%s_%s' % (typ, k)] = v[:, :, ::4, ::4, ::4] if k == 'shapes' else v
        if typ == 'echoscene':
            out['echoscene_z'] = h.last_z
        dec = (h.objs, h.triples, h.tf, h.rf)
        np.random.seed(123)
        keep, d = h.call(lambda: m.sample_boxes_and_shape_with_changes(*dec, *dec, manipulated, **(
            {} if typ == 'echolayout' else {'gen_shape': True})))
        out['%s_chg_keep' % typ] = keep
        out['%s_chg_rel' % typ] = m.diff.LayoutDiff.rel
        for k, v in d.items():
            if v is not None:
                out['%s_chg_%s' % (typ, k)] = v[:, :, ::4, ::4, ::4] if k == 'shapes' else v
        if typ == 'echoscene':
            out['echoscene_chg_z'] = h.last_z
        out.update(objs=h.objs, triples=h.triples)
    out.update(manipulated=np.array(manipulated))
    save('scene_flags_tiny', **out)


def case_sampler_options():
    """The two sampler options the reference's loops take beyond the shipped call (VERDICT r3 "missing #7"):
    ``clip_denoised=True`` of the layout loop (gen_samples_sg -> p_sample_loop_sg -> p_mean_variance clamps the predicted x0 to
    [-1, 1], diffusion_ddpm.py:243-244): all 100 steps at tiny width, injected noise;
    ``eta != 0`` of the DDIM sampler (sigma_t * randn per step and object, samplers/ddim.py:256-260, sigmas from
    make_ddim_sampling_parameters): 4 steps at tiny width with eta = 0.7, the per-step draws injected through noise_like, plus the
    sigma tables of S = 4 and S = 100."""
    net, kw = _unet1d(128, 128)
    fill(net, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    oe, triples, x, _, _ = _layout_loop(net, kw, 8, 3, 100, 100, noise, clip_denoised=True)
    out = dict(layout_obj_embed=oe, layout_triples=triples, layout_x_final_clip=x)
    from model.networks.diffusion_shape.echo2shape import EchoToShape
    from model.networks.diffusion_shape.samplers import ddim as ddim_mod
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    net3 = _unet3d(32, 64)
    fill(net3, 'unet3d_tiny.')
    shim = _ShapeShim()
    shim.df = shim.df_module = net3
    EchoToShape.register_schedule(shim, timesteps=1000, linear_start=0.00085, linear_end=0.012)
    shim.apply_model = lambda *a, **k: EchoToShape.apply_model(shim, *a, **k)
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    O = 4
    objs, tri3 = synth.synthetic_graph(O, seed=6)
    uc, c = rnd((O, 1, 64), 52), rnd((O, 1, 64), 53)
    noise1 = synth.shape_noise(seed=7)
    step_noise = torch.stack([rnd((O, 3, 16, 16, 16), 900 + k) for k in range(4)])
    calls = {'n': 0}
    _orig = ddim_mod.noise_like

    def nl(shape, device, repeat=False):
        k = calls['n']
        calls['n'] += 1
        assert tuple(shape) == (O, 3, 16, 16, 16) and not repeat
        return step_noise[k].clone()
    ddim_mod.noise_like = nl
    try:
        sampler = DDIMSampler(shim)
        with torch.no_grad():
            z, _ = sampler.sample(S=4, batch_size=O, shape=(3, 16, 16, 16), conditioning=c, x_T=noise1.repeat(O, 1, 1, 1, 1),
                                  verbose=False, unconditional_guidance_scale=3., unconditional_conditioning=uc, triplet=tri3, eta=0.7)
    finally:
        ddim_mod.noise_like = _orig
    assert calls['n'] == 4
    s100 = DDIMSampler(shim)
    s100.make_schedule(ddim_num_steps=100, ddim_eta=0.7, verbose=False)
    out.update(ddim_uc_s=uc, ddim_triples=tri3, ddim_z_final_eta07=z, ddim_sigmas_4=torch.as_tensor(np.asarray(sampler.ddim_sigmas)),
               ddim_sigmas_100=torch.as_tensor(np.asarray(s100.ddim_sigmas)))
    save('sampler_options_tiny', **out)


def case_unet1d_mc384():
    """VERDICT r4 "missing #3": GroupNorm32(32, channels) takes every channels % 32 == 0 (ldm_diffusion_util.py:222-239).  A layout
    denoiser with model_channels = 384 has GroupNorm groups of 12 and 24 channels (and 36 behind the skip concatenations) -- none a
    power of two.  Reference modules, seeded weights, both conditioning families: eps at one timestep (O = 8), and 3 ancestral
    steps of a 100-step schedule through the reference's loop."""
    for tag, concat in (('crossattn', False), ('concat', True)):
        net, kw = _unet1d(384, 128, concat=concat)
        fill(net, 'unet1d_mc384_%s.' % tag)
        objs, triples = synth.synthetic_graph(8, seed=31)
        box = rnd((8, 8), 311)
        oe = rnd((8, 640), 312)
        t = torch.full((8,), 437, dtype=torch.int64)
        with torch.no_grad():
            eps = net(box, oe, triples, t)
        noise = synth.layout_noise(8, 8, 100, seed=11)
        oe2, tri2, x, traj, _ = _layout_loop(net, kw, 8, 32, 100, 3, noise)
        save('unet1d_mc384_' + tag, box=box, obj_embed=oe, triples=triples, t=t, eps=eps.squeeze(-1), loop_obj_embed=oe2,
             loop_triples=tri2, loop_x3=x)


def case_sampler_variants():
    """The parameterisations GaussianDiffusion can sample with beyond the shipped 'linear' / 'eps' / 'fixedsmall' (VERDICT r4 "missing
    #4"): the warm-up beta schedules of get_betas (diffusion_ddpm.py:38-58), x0-prediction (:246-254) and the 'fixedlarge' variance
    (:224-235).  All 100 steps at tiny width through the reference's own gen_samples_sg, injected noise; the 'cosine' branch of
    get_betas is recorded as what it does in the reference: it raises."""
    net, kw = _unet1d(128, 128)
    fill(net, 'unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    out = {}
    for tag, ov in (('warm01_large', dict(schedule_type='warm0.1', model_var_type='fixedlarge')),
                    ('warm05_x0', dict(schedule_type='warm0.5', model_mean_type='x0')),
                    ('warm02_x0_large_clip', dict(schedule_type='warm0.2', model_mean_type='x0', model_var_type='fixedlarge'))):
        oe, triples, x, _, tabs = _layout_loop(net, kw, 8, 3, 100, 100, noise, clip_denoised=tag.endswith('clip'), **ov)
        out['x_final_' + tag] = x
        out.update(obj_embed=oe, triples=triples)
    from model.networks.diffusion_layout.diffusion_ddpm import get_betas
    for st in ('warm0.1', 'warm0.2', 'warm0.5'):
        out['betas_' + st.replace('.', '')] = torch.from_numpy(get_betas(st, 1e-4, 0.02, 1000))
    try:
        get_betas('cosine', 1e-4, 0.02, 1000)
        out['cosine_raises'] = torch.zeros(1)
    except UnboundLocalError:
        out['cosine_raises'] = torch.ones(1)
    save('sampler_variants_tiny', **out)


def case_temb():
    """a8: the reference's timestep_embedding (ldm_diffusion_util.py:174-194) for both schedules: t = 999..0 at dim 512
    (layout) and the 100 DDIM timesteps at dim 224 (shape) -- pins the product's host tables bit for bit."""
    from model.networks.diffusion_shape.ldm_diffusion_util import timestep_embedding
    t1 = torch.arange(999, -1, -7, dtype=torch.int64)          # every 7th step (143 rows) keeps the fixture small
    t2 = torch.from_numpy((np.arange(0, 1000, 10) + 1)[::-1].copy())
    save('temb_tables', t_layout=t1, emb_layout=timestep_embedding(t1, 512, repeat_only=False),
         t_shape=t2, emb_shape=timestep_embedding(t2, 224, repeat_only=False))


def case_manifest():
    """checkpoint-key manifest of the reference's own SGDiff built from its REAL shipped YAMLs
    (config/full_mp.yaml, full.yaml, full_concat_mp.yaml -> echoscene; box.yaml, box_no_iou.yaml -> echolayout),
    separated False/True: every state_dict key with its shape for the three checkpoint sections of SURVEY 3.3
    ('diff' module keys, 'shape_df', 'vqvae').  Stored with the parsed config VALUES (numbers / names only, nested
    df_cfg / vq_cfg resolved) so that the consumer can construct its mirror without the reference tree."""
    import gzip
    import json
    import tempfile
    import yaml
    tmp = tempfile.mkdtemp(prefix='golden_manifest_')
    vq = _vqvae(64, 8192)
    vq_path = os.path.join(tmp, 'vq.pth')
    torch.save(vq.state_dict(), vq_path)
    import model.networks.diffusion_shape.echo2shape as e2s
    e2s.init_mesh_renderer = lambda **k: None
    from model.SGDiff import SGDiff
    out = {}
    for cfg_name, typ in (('full_mp', 'echoscene'), ('full', 'echoscene'), ('full_concat_mp', 'echoscene'),
                          ('box', 'echolayout'), ('box_no_iou', 'echolayout')):
        with open(os.path.join('..', 'config', cfg_name + '.yaml')) as f:
            raw = yaml.safe_load(f)
        plain = json.loads(json.dumps(raw))
        if 'shape_branch' in plain:
            for k in ('df_cfg', 'vq_cfg'):
                with open(plain['shape_branch'][k]) as f:
                    plain['shape_branch'][k] = yaml.safe_load(f)
        for separated in (False, True):
            opt = escfg.to_plain(json.loads(json.dumps(plain)))
            opt.hyper.device = 'cpu'
            opt.hyper.logs_dir = opt.hyper.results_dir = tmp
            opt.hyper.isTrain = False
            if 'shape_branch' in opt:
                opt.shape_branch.vq_ckpt = vq_path
            m = SGDiff(typ, opt, synth.VOCAB, replace_latent=False, with_changes=True, residual=True,
                       gconv_pooling='avg', with_angles=True, clip=True, separated=separated)
            ent = {'diff': {k: list(v.shape) for k, v in torch.nn.Module.state_dict(m.diff).items()}}
            if typ == 'echoscene':
                ent['shape_df'] = {k: list(v.shape) for k, v in m.diff.ShapeDiff.df.state_dict().items()}
                ent['vqvae'] = {k: list(v.shape) for k, v in m.diff.ShapeDiff.vqvae.state_dict().items()}
            out['%s|%s|sep%d' % (cfg_name, typ, int(separated))] = ent
            print(cfg_name, typ, separated, {k: len(v) for k, v in ent.items()})
        out['config|' + cfg_name] = plain
    path = os.path.join(HERE, 'key_manifest.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(out, f, sort_keys=True)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def case_scene_e2e_concat():
    """Same as case_scene_e2e for the 'concat' family (config/full_concat_mp.yaml equivalent, tiny widths)."""
    case_scene_e2e(concat=True)


def case_scene_e2e_O2():
    """The smallest scene the dataset can produce: ONE object and the scene node, one triple."""
    case_scene_e2e(concat=False, O=2, name='scene_e2e_O2_tiny')


def case_scene_e2e_norel():
    """``layout_branch.relation_condition: false`` (echo2layout.py:12,105): the reference's ``SGDiff('echolayout')`` end to end.  No
    shipped YAML sets it; the reference then hands ``context=None`` to UNet1DModel.forward, which overwrites it (denoise_net.py:791)."""
    case_scene_e2e(concat=False, name='scene_e2e_norel_tiny', types=('echolayout',), relation_condition=False)


def case_scene_e2e(concat=False, O=8, name=None, types=None, relation_condition=True):
    """The full boundary: the reference's ``SGDiff`` API end to end on CPU (SURVEY.md section 8(c)
    recipe) with a tiny-width config -- setup GCNs, 100-step layout loop, 4-step DDIM, VQ-VAE decode."""
    import tempfile
    import random
    tmp = tempfile.mkdtemp(prefix='golden_e2e_')
    # tiny VQ-VAE checkpoint file required at construction (model/model_utils.py:21)
    vq = _vqvae(32, 64)
    fill(vq, 'e2e.vqvae.')
    vq_path = os.path.join(tmp, 'vq.pth')
    torch.save(vq.state_dict(), vq_path)
    opt = escfg.tiny_diff_opt(device='cpu', logs_dir=tmp, vq_ckpt=vq_path, concat=concat)
    opt.misc.debug = 0
    opt.layout_branch.relation_condition = relation_condition
    import model.networks.diffusion_shape.echo2shape as e2s
    e2s.init_mesh_renderer = lambda **k: None
    from model.networks.diffusion_shape.samplers.ddim import DDIMSampler
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    from model.SGDiff import SGDiff
    out = {}
    for typ in (types or (('echoscene',) if concat else ('echoscene', 'echolayout'))):
        m = SGDiff(typ, opt, synth.VOCAB, replace_latent=False, with_changes=True, residual=True,
                   gconv_pooling='avg', with_angles=True, clip=True, separated=False)
        synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), seed=0, prefix='e2e.diff.')
        if typ == 'echoscene':
            fill(m.diff.ShapeDiff.df, 'e2e.shape_df.')
            m.diff.ShapeDiff.ddim_steps = 4
        m.eval()
        objs, triples = synth.synthetic_graph(O, seed=9)
        tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
        noise = synth.layout_noise(O, 8, 100, seed=7)
        noise1 = synth.shape_noise(seed=7)
        calls = {'n': 0}

        # inject both loops' noise: the layout loop's ``noise_fn=torch.randn`` default is bound at import time,
        # so it is overridden on the reference's own DiffusionPoint.gen_samples_sg; rel2shape's single shared
        # latent noise is drawn through the global torch.randn at call time.
        import model.networks.diffusion_layout.diffusion_ddpm as dd
        _orig_gen = dd.DiffusionPoint.gen_samples_sg

        def noise_fn(size, dtype, device):
            i = calls['n']
            calls['n'] += 1
            return noise[i].clone()

        def gen(self, shape, device, obj_embed, triples=None, condition=None, noise_fn_=None, clip_denoised=True,
                keep_running=False, **kw):
            return _orig_gen(self, shape, device, obj_embed, triples, condition=condition, noise_fn=noise_fn,
                             clip_denoised=clip_denoised, keep_running=keep_running)
        dd.DiffusionPoint.gen_samples_sg = gen
        _randn = torch.randn

        def randn(*a, **k):
            size = k.get('size', a[0] if len(a) == 1 and not isinstance(a[0], int) else a)
            if tuple(size) == (1, 3, 16, 16, 16):
                return noise1.clone()
            return _randn(*a, **k)
        torch.randn = randn
        try:
            with torch.no_grad():
                d = m.sample_box_and_shape(objs, triples, tf, rf, gen_shape=(typ == 'echoscene'))
        finally:
            torch.randn = _randn
            dd.DiffusionPoint.gen_samples_sg = _orig_gen
        assert calls['n'] == 101, calls
        for k, v in d.items():
            if v is None:
                continue
            out['%s_%s' % (typ, k)] = v[:, :, ::4, ::4, ::4] if k == 'shapes' else v
            if k == 'shapes':
                out['%s_shapes_abs' % typ] = v.double().abs().sum()
    out.update(objs=objs, triples=triples)
    save(name or ('scene_e2e_concat_tiny' if concat else 'scene_e2e_tiny'), **out)


def case_gcn_ragged():
    """GraphTripleConvNet on a ragged graph: a node without any triple (avg pooling divides by the clamped count, model/graph.py:
    185-191), a hub that is subject or object of most triples, a repeated (s, o) pair with two predicates and a self-loop; and the
    degenerate graph of ONE node with one self-loop triple."""
    from model.graph import GraphTripleConvNet
    tri = [[0, 1, 1], [0, 2, 2], [0, 3, 4], [0, 4, 5], [0, 5, 6], [2, 3, 0], [4, 1, 0], [1, 2, 2], [1, 5, 2], [6, 3, 6],
           [5, 2, 4], [7, 1, 0], [0, 6, 7]]                       # node 3 has no triple; (1, 2) twice; (6, 6) is a self-loop
    for tag, triples, O in (('ragged', torch.tensor(tri, dtype=torch.int64), 8), ('one_node', torch.tensor([[0, 1, 0]], dtype=torch.int64), 1)):
        net = GraphTripleConvNet(input_dim_obj=96, input_dim_pred=32, num_layers=3, hidden_dim=64,
                                 residual=True, pooling='avg', mlp_normalization='batch', output_dim=80)
        fill(net, 'gcn_res_bn.')                                   # the weights of the 'res_bn' case
        obj = rnd((O, 96), 21)
        pred = rnd((triples.shape[0], 32), 22)
        edges = torch.stack([triples[:, 0], triples[:, 2]], 1)
        with torch.no_grad():
            o, p = net(obj, pred, edges)
        save('gcn_' + tag, obj=obj, pred=pred, triples=triples, out_obj=o, out_pred=p,
             cfg=np.array([96, 32, 3, 64, 1, 1, 80]))


def case_gcn_pooling():
    """GraphTripleConvNet with the two poolings no shipped config selects (model/graph.py:105): 'sum' and the learned 'wAvg'
    (WeightNetGCN, graph.py:37-86, 163-184 -- its down_sample_pred is built for output_dim inputs and fed predicate vectors, so
    the net only runs when output_dim == input_dim_pred at every layer), on the 8-node synthetic graph and on the ragged graph
    of case_gcn_ragged (a node without triples: 0 / (0 + 1e-4))."""
    from model.graph import GraphTripleConvNet
    tri = [[0, 1, 1], [0, 2, 2], [0, 3, 4], [0, 4, 5], [0, 5, 6], [2, 3, 0], [4, 1, 0], [1, 2, 2], [1, 5, 2], [6, 3, 6],
           [5, 2, 4], [7, 1, 0], [0, 6, 7]]
    _, tri8 = synth.synthetic_graph(8, seed=1)
    graphs = dict(g8=tri8, ragged=torch.tensor(tri, dtype=torch.int64))
    for pool, code, (din, dp, H, dout) in (('sum', 1, (96, 32, 64, 80)), ('wAvg', 2, (64, 64, 96, 64))):
        net = GraphTripleConvNet(input_dim_obj=din, input_dim_pred=dp, num_layers=3, hidden_dim=H,
                                 residual=True, pooling=pool, mlp_normalization='batch', output_dim=dout)
        fill(net, 'gcn_%s.' % pool)
        for gname, triples in graphs.items():
            obj = rnd((8, din), 31)
            pred = rnd((triples.shape[0], dp), 32)
            edges = torch.stack([triples[:, 0], triples[:, 2]], 1)
            with torch.no_grad():
                o, p = net(obj, pred, edges)
            save('gcn_%s_%s' % (pool.lower(), gname), obj=obj, pred=pred, triples=triples, out_obj=o, out_pred=p,
                 cfg=np.array([din, dp, 3, H, 1, 1, dout, code]))


CASES = dict(unet1d_mc384=case_unet1d_mc384, sampler_variants=case_sampler_variants, gcn_pooling=case_gcn_pooling, sampler_options=case_sampler_options, unet1d_no_temb=case_unet1d_no_temb, scene_flags=case_scene_flags, gcn_ragged=case_gcn_ragged, scene_e2e_O2=case_scene_e2e_O2, box_post=case_box_post, nomp=case_nomp, concat=case_concat, gcn=case_gcn, unet1d_tiny=case_unet1d_tiny, layout_loop_tiny=case_layout_loop_tiny,
             ddpm_tables=case_ddpm_tables, unet1d_full=case_unet1d_full, unet3d_tiny=case_unet3d_tiny,
             ddim_tiny=case_ddim_tiny, unet3d_full=case_unet3d_full, vqvae=case_vqvae,
             scene_e2e=case_scene_e2e, scene_e2e_concat=case_scene_e2e_concat, scene_e2e_norel=case_scene_e2e_norel,
             layout_traj_full=case_layout_traj_full, shape_traj_full=case_shape_traj_full,
             scene_edit=case_scene_edit, temb=case_temb, manifest=case_manifest,
             unet3d_full_O32=case_unet3d_full_O32, shape_traj_full_O16=case_shape_traj_full_O16)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--ref', default='/root/reference')
    a = ap.parse_args()
    install_reference(a.ref)
    os.chdir(os.path.join(a.ref, 'scripts'))
    torch.manual_seed(0)
    for name, fn in CASES.items():
        if a.only and name not in a.only.split(','):
            continue
        print('== ' + name)
        fn()
