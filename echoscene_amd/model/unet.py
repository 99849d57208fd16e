"""Denoiser parameter trees: the 1-D box UNet and the 3-D latent-SDF UNet.

Mirrors the key layout of the reference's ``UNet1DModel``
(model/networks/diffusion_layout/denoise_net.py:451-740) and ``UNet3DModel``
(model/networks/diffusion_shape/openai_model_3d.py:452-782); both are the same
OpenAI-UNet skeleton, so one table-driven builder serves both.  ``topology()``
is the single description of the block sequence; the holders below and the HIP
plan compiler (echoscene_amd/plan.py) both walk it.
"""
import torch.nn as nn
from .params import _Holder, Lin, Conv, Affine, Emb, Slot, seq
from .graph import GraphTripleConvNet


def topology(model_channels, channel_mult, num_res_blocks, attention_resolutions):
    """Block sequence of the OpenAI UNet as used by both denoisers.

    Returns (input_blocks, middle, output_blocks); each block is a list of
    ('conv_in',) / ('res', cin, cout) / ('attn', ch) / ('down', ch) / ('up', ch).
    Follows the constructor loops at denoise_net.py:553-700 /
    openai_model_3d.py:566-720 (resblock_updown=False, conv_resample=True).
    """
    inp = [[('conv_in',)]]
    chans = [model_channels]
    ch, ds = model_channels, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            blk = [('res', ch, mult * model_channels)]
            ch = mult * model_channels
            if ds in attention_resolutions:
                blk.append(('attn', ch))
            inp.append(blk)
            chans.append(ch)
        if level != len(channel_mult) - 1:
            inp.append([('down', ch)])
            chans.append(ch)
            ds *= 2
    mid = [('res', ch, ch), ('attn', ch), ('res', ch, ch)]
    out = []
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            blk = [('res', ch + ich, model_channels * mult)]
            ch = model_channels * mult
            if ds in attention_resolutions:
                blk.append(('attn', ch))
            if level and i == num_res_blocks:
                blk.append(('up', ch))
                ds //= 2
            out.append(blk)
    return inp, mid, out


class ResBlock(_Holder):
    def __init__(self, dims, cin, emb_ch, cout):
        super().__init__()
        self.in_layers = seq(Affine(cin), Slot(), Conv(dims, cin, cout, 3))
        self.emb_layers = seq(Slot(), Lin(emb_ch, cout))
        self.out_layers = seq(Affine(cout), Slot(), Slot(), Conv(dims, cout, cout, 3))
        self.skip_connection = Slot() if cin == cout else Conv(dims, cin, cout, 1)


class CrossAttention(_Holder):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.to_q = Lin(query_dim, inner, bias=False)
        self.to_k = Lin(context_dim, inner, bias=False)
        self.to_v = Lin(context_dim, inner, bias=False)
        self.to_out = seq(Lin(inner, query_dim), Slot())


class GEGLU(_Holder):
    def __init__(self, din, dout):
        super().__init__()
        self.proj = Lin(din, dout * 2)


class FeedForward(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.net = seq(GEGLU(dim, dim * 4), Slot(), Lin(dim * 4, dim))


class BasicTransformerBlock(_Holder):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm1, self.norm2, self.norm3 = Affine(dim), Affine(dim), Affine(dim)


class SpatialTransformer(_Holder):
    """SpatialTransformer1D / 3D (attention.py:298-396): GN, 1x1 proj_in, depth-1
    transformer, 1x1 proj_out."""

    def __init__(self, dims, ch, heads, dim_head, context_dim):
        super().__init__()
        inner = heads * dim_head
        self.norm = Affine(ch)
        self.proj_in = Conv(dims, ch, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, context_dim)])
        self.proj_out = Conv(dims, inner, ch, 1)


class AttentionBlock(_Holder):
    """AttentionBlock of the 'concat' family (denoise_net.py:316-363 == openai_model_3d.py:317-363): GroupNorm32,
    Conv1d qkv with rows ordered [head][q|k|v][ch] (QKVAttentionLegacy), Conv1d proj_out."""

    def __init__(self, ch):
        super().__init__()
        self.norm = Affine(ch)
        self.qkv = Conv(1, ch, 3 * ch, 1)
        self.proj_out = Conv(1, ch, ch, 1)


class Downsample(_Holder):
    def __init__(self, dims, ch):
        super().__init__()
        self.op = Conv(dims, ch, ch, 3)


class Upsample(_Holder):
    def __init__(self, dims, ch):
        super().__init__()
        self.conv = Conv(dims, ch, ch, 3)


class _UNetTrunk(_Holder):
    def _build_trunk(self, dims, in_channels, out_channels, model_channels, channel_mult,
                     num_res_blocks, attention_resolutions, num_heads, context_dim, transformer=True):
        emb = model_channels * 4
        self.time_embed = seq(Lin(model_channels, emb), Slot(), Lin(emb, emb))
        inp, mid, out = topology(model_channels, list(channel_mult), num_res_blocks,
                                 list(attention_resolutions))
        self.topo = (inp, mid, out)

        def make(item):
            kind = item[0]
            if kind == 'conv_in':
                return Conv(dims, in_channels, model_channels, 3)
            if kind == 'res':
                return ResBlock(dims, item[1], emb, item[2])
            if kind == 'attn':
                if not transformer:
                    return AttentionBlock(item[1])
                return SpatialTransformer(dims, item[1], num_heads, item[1] // num_heads, context_dim)
            if kind == 'down':
                return Downsample(dims, item[1])
            if kind == 'up':
                return Upsample(dims, item[1])
            raise ValueError(kind)

        self.input_blocks = nn.ModuleList([seq(*[make(i) for i in blk]) for blk in inp])
        self.middle_block = seq(*[make(i) for i in mid])
        self.output_blocks = nn.ModuleList([seq(*[make(i) for i in blk]) for blk in out])
        self.out = seq(Affine(model_channels), Slot(), Conv(dims, model_channels, out_channels, 3))


class UNet1DModel(_UNetTrunk):
    """Layout denoiser.  Constructor keywords are those of
    ``config/*.yaml: layout_branch.denoiser_kwargs`` (denoise_net.py:480-506)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True,
                 dims=1, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 concat_dim=None, crossattn_dim=None, conditioning_key='crossattn', using_clip=True,
                 enable_t_emb=False):
        super().__init__()
        self.concat = conditioning_key == 'concat'
        if not ((conditioning_key == 'crossattn' and use_spatial_transformer and transformer_depth == 1) or
                (self.concat and not use_spatial_transformer)):
            raise NotImplementedError(
                "layout denoiser: 'crossattn' + spatial transformer (depth 1, config/full_mp.yaml) or 'concat' + "
                "AttentionBlock (config/full_concat_mp.yaml)")
        if dims != 1 or use_scale_shift_norm or resblock_updown or num_heads == -1 or num_head_channels != -1 \
                or use_new_attention_order:
            raise NotImplementedError("unsupported UNet1DModel option")
        # NOTE: the reference stores in_channels + concat_dim in self.in_channels for 'concat' (denoise_net.py:531);
        # here in_channels stays the box dimension and trunk_in_channels is the input conv's.
        self.dims, self.in_channels, self.out_channels = 1, in_channels, out_channels
        self.trunk_in_channels = in_channels + (concat_dim if self.concat else 0)
        self.model_channels, self.num_heads = model_channels, num_heads
        self.conditioning_key, self.using_clip, self.enable_t_emb = conditioning_key, using_clip, enable_t_emb
        self.context_dim = None if self.concat else crossattn_dim
        self._build_trunk(1, self.trunk_in_channels, out_channels, model_channels, channel_mult,
                          num_res_blocks, attention_resolutions, num_heads, self.context_dim,
                          transformer=not self.concat)
        g = 64  # gconv_dim hard-coded by the reference (denoise_net.py:717)
        add = 512 if using_clip else 0
        self.pred_embeddings = Emb(16, 2 * g)
        self.box_embeddings = Lin(in_channels, g)
        obj_dim = 2 * g + add + g
        if enable_t_emb:
            self.box_time_emb = Lin(model_channels * 4, g)
            obj_dim += g
        self.box_graph_cov = GraphTripleConvNet(
            input_dim_obj=obj_dim, input_dim_pred=2 * g, hidden_dim=4 * g, pooling='avg',
            num_layers=5, mlp_normalization='batch', residual=True, output_dim=concat_dim)


class UNet3DModel(_UNetTrunk):
    """Latent-SDF denoiser (openai_model_3d.py:480-782); keywords are those of
    ``config/sdfusion-txt2shape_mp.yaml: unet.params``."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True,
                 dims=2, num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=False,
                 transformer_depth=1, context_dim=None, n_embed=None, legacy=True, using_clip=True,
                 messsage_passing=False, enable_t_emb=False, conditioning_key='concat'):
        super().__init__()
        self.concat = conditioning_key == 'concat'
        if not ((conditioning_key == 'crossattn' and use_spatial_transformer and transformer_depth == 1 and dims == 3) or
                (self.concat and not use_spatial_transformer and dims == 4)):
            raise NotImplementedError(
                "shape denoiser: 'crossattn' + spatial transformer, dims 3 (sdfusion-txt2shape_mp.yaml) or 'concat' + "
                "AttentionBlock, dims 4 (sdfusion-txt2shape_concat_mp.yaml)")
        if num_classes is not None or n_embed is not None or num_heads == -1 or num_head_channels != -1 \
                or use_scale_shift_norm or resblock_updown or use_new_attention_order:
            raise NotImplementedError("unsupported UNet3DModel option")
        # dims == 4 is still Conv3d (ldm_diffusion_util.py:251-252) but strides / up-samples ALL three axes
        self.dims, self.in_channels, self.out_channels = dims, in_channels, out_channels
        self.model_channels, self.num_heads, self.image_size = model_channels, num_heads, image_size
        self.conditioning_key, self.messsage_passing, self.enable_t_emb = \
            conditioning_key, messsage_passing, enable_t_emb
        context_dim = 4096 if self.concat else context_dim      # x_dim (openai_model_3d.py:749-754)
        self.context_dim = context_dim
        self._build_trunk(3, in_channels, out_channels, model_channels, channel_mult,
                          num_res_blocks, attention_resolutions, num_heads, None if self.concat else context_dim,
                          transformer=not self.concat)
        if messsage_passing:
            g = 64
            self.pred_embeddings = Emb(16, 2 * g)
            # conv-pool stem that turns x_t into a 64-d shape code (openai_model_3d.py:757-764)
            self.shape_embeddings = nn.ModuleList([
                Conv(3, 4 if self.concat else 3, 32, 3), Slot(), Conv(3, 32, 64, 3), Slot(), Slot(), Lin(64 * 2 * 2 * 2, g)])
            obj_dim = g + context_dim
            if enable_t_emb:
                self.shape_time_emb = Lin(model_channels * 4, g)
                obj_dim += g
            self.shape_code_graph_cov = GraphTripleConvNet(
                input_dim_obj=obj_dim, input_dim_pred=2 * g, hidden_dim=4 * g, pooling='avg',
                num_layers=5, mlp_normalization='batch', residual=True, output_dim=context_dim)


class DiffusionUNet(_Holder):
    """Wrapper whose only job is the ``diffusion_net.`` key prefix
    (diffusion_shape/network.py:11-19)."""

    def __init__(self, unet_params, conditioning_key='crossattn'):
        super().__init__()
        self.conditioning_key = conditioning_key
        kw = dict(unet_params)
        kw['conditioning_key'] = conditioning_key
        self.diffusion_net = UNet3DModel(**kw)
