"""Minimal OmegaConf-compatible config objects.

The reference reads its model hyper-parameters with OmegaConf
(scripts/eval_3dfront.py:383-387, model/networks/diffusion_shape/echo2shape.py:62-63).
omegaconf is not a dependency of this build; ``AttrDict`` gives the same access
patterns the hot path needs (attribute access, ``.get``, ``**`` splatting,
``1e-4`` parsed as float -- SURVEY.md appendix B item 8).  A real OmegaConf
``DictConfig`` passed by ``eval_3dfront.py`` works too: ``to_plain`` converts it.
"""
import os
import re
import yaml

_FLOAT = re.compile(r'^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$')


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [_wrap(v) for v in x]
    if isinstance(x, str) and _FLOAT.match(x):
        return float(x)
    return x


def to_plain(cfg):
    """AttrDict / OmegaConf DictConfig / dict -> AttrDict tree."""
    if isinstance(cfg, AttrDict):
        return cfg
    if isinstance(cfg, dict):
        return _wrap(cfg)
    try:  # OmegaConf object, if the caller has omegaconf installed
        from omegaconf import OmegaConf
        return _wrap(OmegaConf.to_container(cfg, resolve=True))
    except Exception:
        return _wrap(dict(cfg))


def load_yaml(path):
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


# ---------------------------------------------------------------------------
# Built-in defaults = the values of the reference's shipped hot-path configs, so that
# benchmarks and tests do not need /root/reference at run time.
# config/full_mp.yaml:18-52 (layout), config/sdfusion-txt2shape_mp.yaml (shape),
# config/vqvae_snet.yaml (VQ-VAE).  SURVEY.md appendix A.
# ---------------------------------------------------------------------------
def layout_denoiser_kwargs(model_channels=512, enable_t_emb=True, concat=False):
    """config/full_mp.yaml:18-37; ``concat=True``: config/full_concat_mp.yaml:18-37 (AttentionBlocks instead of
    spatial transformers, the GCN output appended to the box vector)."""
    return AttrDict(
        dims=1, in_channels=8, out_channels=8, model_channels=model_channels,
        channel_mult=[1, 1, 1, 1], num_res_blocks=2, attention_resolutions=[4, 2], num_heads=8,
        use_spatial_transformer=not concat, transformer_depth=1,
        conditioning_key='concat' if concat else 'crossattn',
        concat_dim=1280, crossattn_dim=1280, use_checkpoint=True, enable_t_emb=enable_t_emb)


def layout_diffusion_kwargs(time_num=1000):
    return AttrDict(schedule_type='linear', beta_start=0.0001, beta_end=0.02, time_num=time_num,
                    model_mean_type='eps', model_var_type='fixedsmall', loss_separate=True,
                    loss_iou=False, iou_type='obb', train_stats_file=None)


def shape_unet_params(model_channels=224, concat=False, mp=True):
    """config/sdfusion-txt2shape_mp.yaml; ``concat=True``: config/sdfusion-txt2shape_concat_mp.yaml (in_channels 5,
    dims 4 = Conv3d with stride 2 / nearest x2 in all three axes, AttentionBlocks, no context).  ``mp=False``: the
    configs without echo message passing (sdfusion-txt2shape.yaml / sdfusion-txt2shape_concat.yaml, in_channels 3 / 4)."""
    if concat:
        return AttrDict(
            image_size=16, in_channels=5 if mp else 4, out_channels=3, model_channels=model_channels,
            num_res_blocks=2, attention_resolutions=[4, 2], channel_mult=[1, 2, 3], num_heads=8, dims=4,
            use_spatial_transformer=False, transformer_depth=1, context_dim=None, use_checkpoint=True,
            legacy=False, messsage_passing=mp, enable_t_emb=mp)
    return AttrDict(
        image_size=16, in_channels=3, out_channels=3, model_channels=model_channels,
        num_res_blocks=2, attention_resolutions=[4, 2], channel_mult=[1, 2, 3], num_heads=8, dims=3,
        use_spatial_transformer=True, transformer_depth=1, context_dim=1280, use_checkpoint=True,
        legacy=False, messsage_passing=mp, enable_t_emb=mp)


def shape_df_conf(model_channels=224, concat=False, mp=True):
    return AttrDict(
        model=AttrDict(params=AttrDict(linear_start=0.00085, linear_end=0.012,
                                       conditioning_key='concat' if concat else 'crossattn', timesteps=1000,
                                       scale_factor=0.18215)),
        unet=AttrDict(params=shape_unet_params(model_channels, concat, mp)))


def vqvae_conf(ch=64):
    return AttrDict(model=AttrDict(params=AttrDict(
        embed_dim=3, n_embed=8192,
        ddconfig=AttrDict(double_z=False, z_channels=3, resolution=64, in_channels=1, out_ch=1,
                          ch=ch, ch_mult=[1, 2, 4], num_res_blocks=1, attn_resolutions=[],
                          dropout=0.0))))


def default_diff_opt(device='cuda', time_num=1000, logs_dir=None, concat=False):
    """Equivalent of ``OmegaConf.load('config/full_mp.yaml')`` (``concat=True``: ``config/full_concat_mp.yaml``) with
    the two nested yaml files inlined (``df_cfg`` / ``vq_cfg`` may also be paths, as in the reference)."""
    return AttrDict(
        hyper=AttrDict(batch_size=64, gpu_ids=0, logs_dir=logs_dir, results_dir=logs_dir, name='./',
                       isTrain=False, device=device, distributed=0, lr_init=1e-4,
                       lr_step=[35000, 70000, 140000], lr_evo=[5e-5, 1e-5, 5e-6]),
        dataset=AttrDict(res=64, trunc_thres=0.2, ratio=1),
        layout_branch=AttrDict(
            model='diffusion_scene_layout_ddpm', angle_dim=2, denoiser='unet1d',
            relation_condition=True, denoiser_kwargs=layout_denoiser_kwargs(concat=concat),
            diffusion_kwargs=layout_diffusion_kwargs(time_num)),
        shape_branch=AttrDict(
            model='sdfusion-txt2shape_concat_mp' if concat else 'sdfusion-txt2shape_mp', sampling='greedy', ckpt=None,
            df_cfg=shape_df_conf(concat=concat), ddim_steps=100, ddim_eta=0.0, uc_scale=3.0,
            vq_model='vqvae', vq_cfg=vqvae_conf(), vq_dset=None, vq_cat=None, vq_ckpt=None),
        misc=AttrDict(debug=0, seed=111, backend='gloo', local_rank=0))


def resolve_nested(cfg_or_path):
    """``df_cfg`` / ``vq_cfg`` are yaml paths in the reference (relative to scripts/);
    accept a path, an AttrDict or an OmegaConf node."""
    if isinstance(cfg_or_path, (str, os.PathLike)):
        return load_yaml(cfg_or_path)
    return to_plain(cfg_or_path)


def tiny_diff_opt(device='cuda', logs_dir=None, vq_ckpt=None, concat=False):
    """Narrow end-to-end test configuration (same topology as full_mp.yaml, widths 128 / 32 / 32,
    100 layout steps, 64-entry codebook).  Used by tests/golden/make_golden.py (on the reference) and
    by the parity tests (on this build)."""
    opt = default_diff_opt(device=device, time_num=100, logs_dir=logs_dir, concat=concat)
    opt.hyper.isTrain = False
    opt.layout_branch.denoiser_kwargs = layout_denoiser_kwargs(128, concat=concat)
    opt.layout_branch.denoiser_kwargs.concat_dim = 128
    opt.layout_branch.denoiser_kwargs.crossattn_dim = 128
    opt.shape_branch.df_cfg = shape_df_conf(32, concat=concat)
    vqc = vqvae_conf(32)
    vqc.model.params.n_embed = 64
    opt.shape_branch.vq_cfg = vqc
    opt.shape_branch.vq_ckpt = vq_ckpt
    return opt
