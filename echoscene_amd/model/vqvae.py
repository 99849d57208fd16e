"""VQ-VAE parameter tree (reference model/networks/vqvae_networks/network.py:49-78,
vqvae_modules.py:179-374).  The encoder is never run on the sampling path but its keys
are part of the ``'vqvae'`` checkpoint entry, so they are kept for ``load_state_dict``.
"""
import torch.nn as nn
from .params import _Holder, Conv, Affine, Emb


class ResnetBlock(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = Affine(cin)
        self.conv1 = Conv(3, cin, cout, 3)
        self.norm2 = Affine(cout)
        self.conv2 = Conv(3, cout, cout, 3)
        if cin != cout:
            self.nin_shortcut = Conv(3, cin, cout, 1)


class AttnBlock(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.norm = Affine(ch)
        self.q, self.k, self.v = Conv(3, ch, ch, 1), Conv(3, ch, ch, 1), Conv(3, ch, ch, 1)
        self.proj_out = Conv(3, ch, ch, 1)


class _Level(_Holder):
    pass


class _Resample(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.conv = Conv(3, ch, ch, 3)


def _mid(ch):
    m = _Level()
    m.block_1, m.attn_1, m.block_2 = ResnetBlock(ch, ch), AttnBlock(ch), ResnetBlock(ch, ch)
    return m


class Encoder3D(_Holder):
    def __init__(self, ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution,
                 z_channels, double_z=True, **_):
        super().__init__()
        self.conv_in = Conv(3, in_channels, ch, 3)
        self.down = nn.ModuleList()
        in_mult = (1,) + tuple(ch_mult)
        res = resolution
        for lvl in range(len(ch_mult)):
            bin_, bout = ch * in_mult[lvl], ch * ch_mult[lvl]
            d = _Level()
            d.block, d.attn = nn.ModuleList(), nn.ModuleList()
            for _b in range(num_res_blocks):
                d.block.append(ResnetBlock(bin_, bout))
                bin_ = bout
                if res in attn_resolutions:
                    d.attn.append(AttnBlock(bin_))
            if lvl != len(ch_mult) - 1:
                d.downsample = _Resample(bin_)
                res //= 2
            self.down.append(d)
        self.mid = _mid(bin_)
        self.norm_out = Affine(bin_)
        self.conv_out = Conv(3, bin_, 2 * z_channels if double_z else z_channels, 3)


class Decoder3D(_Holder):
    def __init__(self, ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels,
                 resolution, z_channels, **_):
        super().__init__()
        n = len(ch_mult)
        bin_ = ch * ch_mult[n - 1]
        res = resolution // 2 ** (n - 1)
        self.conv_in = Conv(3, z_channels, bin_, 3)
        self.mid = _mid(bin_)
        ups = []
        for lvl in reversed(range(n)):
            bout = ch * ch_mult[lvl]
            u = _Level()
            u.block, u.attn = nn.ModuleList(), nn.ModuleList()
            for _b in range(num_res_blocks):
                u.block.append(ResnetBlock(bin_, bout))
                bin_ = bout
                if res in attn_resolutions:
                    u.attn.append(AttnBlock(bin_))
            if lvl != 0:
                u.upsample = _Resample(bin_)
                res *= 2
            ups.insert(0, u)
        self.up = nn.ModuleList(ups)
        self.norm_out = Affine(bin_)
        self.conv_out = Conv(3, bin_, out_ch, 3)


class VectorQuantizer(_Holder):
    def __init__(self, n_e, e_dim):
        super().__init__()
        self.embedding = Emb(n_e, e_dim)


class VQVAE(_Holder):
    def __init__(self, ddconfig, n_embed, embed_dim):
        super().__init__()
        dd = dict(ddconfig)
        self.ddconfig, self.n_embed, self.embed_dim = dd, n_embed, embed_dim
        self.encoder = Encoder3D(**dd)
        self.decoder = Decoder3D(**dd)
        self.quantize = VectorQuantizer(n_embed, embed_dim)
        self.quant_conv = Conv(3, dd['z_channels'], embed_dim, 1)
        self.post_quant_conv = Conv(3, embed_dim, dd['z_channels'], 1)
