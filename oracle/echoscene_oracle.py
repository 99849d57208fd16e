"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A functional, PyTorch-CPU (fp32) restatement of EchoScene's scene-graph diffusion
*sampling* path: the GraphTripleConv message passing, the 1-D box denoiser, the 3-D
latent-SDF denoiser, the DDPM / DDIM loops, the one-off setup GCNs and the VQ-VAE
decode epilogue.  Each function cites the reference file:line it follows
(paths relative to the upstream repo ymxlzgy/echoscene).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module -- as the checker / the timed CPU baseline, never
as the thing shipped.  The product path (echoscene_amd/*) must not import it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the *reference itself*
in the build container, fills its modules with ``echoscene_amd.synth.seeded_tensor``
weights and stores inputs/outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those vectors.
(The reference has no tests or golden vectors of its own -- SURVEY.md section 4.)
Stated here as required: ``chamfer_forward/backward`` restate a CUDA extension (chamfer.cu) that cannot be
built or run in this image (no nvcc, ATen-CUDA), so no output of the reference itself exists for them.  They are pinned the
only way that does not need it: integer-exact lattice fixtures (tests/golden/make_chamfer_lattice.py, ground truth in int64)
on which every fp32 operation of the reference's formula is exact and only its comparison rule -- first minimum, read off
chamfer.cu:37-67,126-130 -- decides the result (tests/test_boundary_cpu.py).  ``descale_box_params`` / ``sincos2arctan`` are
pinned like the rest (golden ``box_post`` from the reference's helpers/util.py).

Everything is driven by a flat ``state_dict`` (name -> tensor) with the reference's
key names, so the network topology is *inferred from the keys* -- independently of
the product's topology table (echoscene_amd/model/unet.py).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# numerics hook: optional emulation of the product's fp16-MFMA operand rounding
# --------------------------------------------------------------------------------------
class Numerics:
    """``trunk_dtype=None``: exact fp32 restatement of the reference.
    ``trunk_dtype=torch.float16``: round both operands of every contraction in the 3-D UNet
    trunk (conv / linear / attention bmm) to fp16 and accumulate in fp32 -- the contract of
    the HIP MFMA path (DESIGN.md section 'numerics').  Used to separate "kernel bug" from
    "precision of the chosen MFMA dtype" in the parity tests."""

    def __init__(self, trunk_dtype=None):
        self.trunk_dtype = trunk_dtype

    def r(self, t):
        return t if self.trunk_dtype is None else t.to(self.trunk_dtype).float()


EXACT = Numerics(None)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _linear(sd, p, x, nm=EXACT, trunk=False):
    w, b = sd[p + '.weight'], sd.get(p + '.bias')
    if trunk:
        return F.linear(nm.r(x), nm.r(w), b)
    return F.linear(x, w, b)


# --------------------------------------------------------------------------------------
# a3  build_mlp  (model/layers.py:21-38)  -- eval-mode BatchNorm1d
# --------------------------------------------------------------------------------------
def mlp(sd, p, x, final_nonlinearity=True):
    idx = sorted({int(k[len(p) + 1:].split('.')[0]) for k in sd if k.startswith(p + '.')})
    lin_idx = [i for i in idx if sd[f'{p}.{i}.weight'].dim() == 2]
    for n, i in enumerate(lin_idx):
        x = _linear(sd, f'{p}.{i}', x)
        last = n == len(lin_idx) - 1
        if not last or final_nonlinearity:
            if f'{p}.{i + 1}.running_mean' in sd:  # BatchNorm1d, running statistics
                q = f'{p}.{i + 1}'
                x = F.batch_norm(x, sd[q + '.running_mean'], sd[q + '.running_var'],
                                 sd[q + '.weight'], sd[q + '.bias'], training=False, eps=1e-5)
            x = F.relu(x)
    return x


# --------------------------------------------------------------------------------------
# a1/a2  GraphTripleConv / GraphTripleConvNet  (model/graph.py:124-211, 246-250)
# --------------------------------------------------------------------------------------
def weight_net_gcn(sd, q, s, p, o):
    """WeightNetGCN.forward (model/graph.py:66-86): a weight in (0, 1) per triple for its subject and its object slot."""
    s = _linear(sd, q + '.down_sample_obj', s)                        # :68
    p = _linear(sd, q + '.down_sample_pred', p)                       # :69
    o = _linear(sd, q + '.down_sample_obj', o)                        # :70
    feat = torch.cat([s, o, p], 1)                                    # :73 / :76 / :79

    def head(h):
        return torch.sigmoid(_linear(sd, f'{q}.{h}.2', F.relu(_linear(sd, f'{q}.{h}.0', feat))))   # :43-48
    if (q + '.Net_s.0.weight') in sd:                                 # separate_s_o=True (the constructor's default)
        return head('Net_s'), head('Net_o')
    w = head('Net')
    return w, w


def graph_triple_conv(sd, p, obj, pred, edges, pooling='avg'):
    assert pooling in ('sum', 'avg', 'wAvg')                          # graph.py:105
    O, T = obj.shape[0], pred.shape[0]
    H = sd[p + '.net2.0.weight'].shape[1]
    Dp = pred.shape[1]
    s_idx, o_idx = edges[:, 0].contiguous(), edges[:, 1].contiguous()
    t_in = torch.cat([obj[s_idx], pred, obj[o_idx]], dim=1)          # graph.py:146-151
    t_out = mlp(sd, p + '.net1', t_in)                                # :152
    new_s, new_p, new_o = t_out[:, :H], t_out[:, H:H + Dp], t_out[:, H + Dp:]   # :156-158
    pooled = torch.zeros(O, H, dtype=obj.dtype)
    if pooling == 'wAvg':                                                       # :163-170
        w_s, w_o = weight_net_gcn(sd, p + '.weightNet', new_s, new_p, new_o)
        new_s = w_s * new_s
        new_o = w_o * new_o
    pooled = pooled.scatter_add(0, s_idx.view(-1, 1).expand_as(new_s), new_s)   # :176
    pooled = pooled.scatter_add(0, o_idx.view(-1, 1).expand_as(new_o), new_o)   # :177
    if pooling == 'wAvg':                                                       # :179-184
        wsum = torch.zeros(O, 1, dtype=obj.dtype)
        wsum = wsum.scatter_add(0, o_idx.view(-1, 1), w_o).scatter_add(0, s_idx.view(-1, 1), w_s)
        pooled = pooled / (wsum + 0.0001)
    if pooling == 'avg':
        counts = torch.zeros(O, dtype=obj.dtype)
        ones = torch.ones(T, dtype=obj.dtype)
        counts = counts.scatter_add(0, s_idx, ones).scatter_add(0, o_idx, ones)     # :189-192
        pooled = pooled / counts.clamp(min=1).view(-1, 1)                           # :198-199
    new_obj = mlp(sd, p + '.net2', pooled)                                      # :203
    if (p + '.linear_projection.weight') in sd:                                # residual, :205-209
        new_obj = new_obj + _linear(sd, p + '.linear_projection', obj)
        new_p = new_p + _linear(sd, p + '.linear_projection_pred', pred)
    return new_obj, new_p


def gcn_net(sd, p, obj, pred, edges, pooling='avg'):
    n = 1 + max(int(k[len(p) + 8:].split('.')[0]) for k in sd if k.startswith(p + '.gconvs.'))
    for i in range(n):
        obj, pred = graph_triple_conv(sd, f'{p}.gconvs.{i}', obj, pred, edges, pooling)
    return obj, pred


def _edges(triples):
    return torch.stack([triples[:, 0], triples[:, 2]], dim=1), triples[:, 1]


# --------------------------------------------------------------------------------------
# a8  timestep_embedding  (diffusion_shape/ldm_diffusion_util.py:174-194)
# --------------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# --------------------------------------------------------------------------------------
# a6/a7  ResBlock, SpatialTransformer, BasicTransformerBlock, CrossAttention, GEGLU
# --------------------------------------------------------------------------------------
def _conv(sd, p, x, nm, trunk, stride=1, padding=1):
    w, b = sd[p + '.weight'], sd.get(p + '.bias')
    if trunk:
        x, w = nm.r(x), nm.r(w)
    if w.dim() == 3:
        return F.conv1d(x, w, b, stride=stride, padding=padding)
    return F.conv3d(x, w, b, stride=stride, padding=padding)


def res_block(sd, p, x, emb, nm=EXACT, trunk=False):
    """denoise_net.py:293-313 == openai_model_3d.py:294-314 (no scale-shift, dropout 0)."""
    h = F.silu(F.group_norm(x, 32, sd[p + '.in_layers.0.weight'], sd[p + '.in_layers.0.bias'], 1e-5))
    h = _conv(sd, p + '.in_layers.2', h, nm, trunk)
    e = _linear(sd, p + '.emb_layers.1', F.silu(emb))
    while e.dim() < h.dim():
        e = e[..., None]
    h = h + e
    h = F.silu(F.group_norm(h, 32, sd[p + '.out_layers.0.weight'], sd[p + '.out_layers.0.bias'], 1e-5))
    h = _conv(sd, p + '.out_layers.3', h, nm, trunk)
    if (p + '.skip_connection.weight') in sd:
        x = _conv(sd, p + '.skip_connection', x, nm, trunk, padding=0)
    return x + h


def cross_attention(sd, p, x, context, heads, nm=EXACT, trunk=False):
    """attention.py:172-219 (without the pdb NaN traps)."""
    ctx = x if context is None else context
    q = _linear(sd, p + '.to_q', x, nm, trunk)
    k = _linear(sd, p + '.to_k', ctx, nm, trunk)
    v = _linear(sd, p + '.to_v', ctx, nm, trunk)
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    if trunk:
        q, k, v = nm.r(q), nm.r(k), nm.r(v)
    sim = torch.einsum('bid,bjd->bij', q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    if trunk:
        attn = nm.r(attn)
    out = torch.einsum('bij,bjd->bid', attn, v)
    out = out.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    return _linear(sd, p + '.to_out.0', out, nm, trunk)


def transformer_block(sd, p, x, context, heads, nm=EXACT, trunk=False, trace=None):
    """attention.py:237-245: self-attn, cross-attn, GEGLU feed-forward, all pre-LN residual."""
    C = x.shape[-1]

    def ln(i, t):
        return F.layer_norm(t, (C,), sd[f'{p}.norm{i}.weight'], sd[f'{p}.norm{i}.bias'], 1e-5)

    if trace is not None:
        trace[p + ':in'] = x
    x = cross_attention(sd, p + '.attn1', ln(1, x), None, heads, nm, trunk) + x
    if trace is not None:
        trace[p + ':attn1'] = x
    x = cross_attention(sd, p + '.attn2', ln(2, x), context, heads, nm, trunk) + x
    if trace is not None:
        trace[p + ':attn2'] = x
    h = _linear(sd, p + '.ff.net.0.proj', ln(3, x), nm, trunk)
    a, gate = h.chunk(2, dim=-1)                                   # attention.py:44-46
    h = a * F.gelu(gate)
    x = _linear(sd, p + '.ff.net.2', h, nm, trunk) + x
    if trace is not None:
        trace[p + ':ff'] = x
    return x


def spatial_transformer(sd, p, x, context, heads, nm=EXACT, trunk=False, trace=None):
    """attention.py:335-351 (3-D) / :385-396 (1-D).  GroupNorm eps is 1e-6 here."""
    shp = x.shape
    h = F.group_norm(x, 32, sd[p + '.norm.weight'], sd[p + '.norm.bias'], 1e-6)
    h = _conv(sd, p + '.proj_in', h, nm, trunk, padding=0)
    B, C = h.shape[:2]
    h = h.reshape(B, C, -1).permute(0, 2, 1)
    h = transformer_block(sd, p + '.transformer_blocks.0', h, context, heads, nm, trunk, trace)
    h = h.permute(0, 2, 1).reshape(B, C, *shp[2:])
    h = _conv(sd, p + '.proj_out', h, nm, trunk, padding=0)
    return h + x


def attention_block(sd, p, x, heads, nm=EXACT, trunk=False):
    """AttentionBlock + QKVAttentionLegacy of the 'concat' model family (denoise_net.py:316-410 ==
    openai_model_3d.py:317-411): GroupNorm32, 1x1 qkv with rows ordered [head][q|k|v][ch], softmax(q k^T / sqrt(ch)) v
    per head, 1x1 proj_out, residual."""
    shp = x.shape
    B, C = shp[:2]
    xf = x.reshape(B, C, -1)
    hn = F.group_norm(xf, 32, sd[p + '.norm.weight'], sd[p + '.norm.bias'], 1e-5)
    w, b = sd[p + '.qkv.weight'], sd[p + '.qkv.bias']
    if trunk:
        hn, w = nm.r(hn), nm.r(w)
    qkv = F.conv1d(hn, w, b)
    T = qkv.shape[-1]
    ch = C // heads
    q, k, v = qkv.reshape(B * heads, 3 * ch, T).split(ch, dim=1)
    if trunk:
        q, k, v = nm.r(q), nm.r(k), nm.r(v)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    wgt = torch.einsum('bct,bcs->bts', q * scale, k * scale).softmax(dim=-1)
    if trunk:
        wgt = nm.r(wgt)
    a = torch.einsum('bts,bcs->bct', wgt, v).reshape(B, C, T)
    wp = sd[p + '.proj_out.weight']
    if trunk:
        a, wp = nm.r(a), nm.r(wp)
    h = F.conv1d(a, wp, sd[p + '.proj_out.bias'])
    return (xf + h).reshape(shp)


def _run_block(sd, p, h, emb, context, heads, nm, trunk, trace=None, full3d=False):
    """One TimestepEmbedSequential; sub-module kinds inferred from the keys.  ``full3d``: the 'concat' shape UNet
    is built with dims=4 (config/sdfusion-txt2shape_concat_mp.yaml), i.e. Conv3d with stride 2 / nearest x2 in ALL
    three axes (openai_model_3d.py:150-156,188) instead of the (1,2,2) of dims=3."""
    j = 0
    while any(k.startswith(f'{p}.{j}.') for k in sd):
        q = f'{p}.{j}'
        if (q + '.in_layers.0.weight') in sd:
            h = res_block(sd, q, h, emb, nm, trunk)
        elif (q + '.transformer_blocks.0.norm1.weight') in sd:
            h = spatial_transformer(sd, q, h, context, heads, nm, trunk, trace)
        elif (q + '.qkv.weight') in sd:
            h = attention_block(sd, q, h, heads, nm, trunk)
        elif (q + '.op.weight') in sd:     # Downsample: stride 2 (1-D, dims=4) or (1,2,2) (dims=3)
            stride = 2 if (sd[q + '.op.weight'].dim() == 3 or full3d) else (1, 2, 2)
            h = _conv(sd, q + '.op', h, nm, trunk, stride=stride)
        elif (q + '.conv.weight') in sd:   # Upsample: nearest, then conv
            if h.dim() == 3:
                h = F.interpolate(h, scale_factor=1, mode='nearest')       # denoise_net.py:154
            elif full3d:
                h = F.interpolate(h, scale_factor=2, mode='nearest')
            else:
                h = F.interpolate(h, (h.shape[2], h.shape[3] * 2, h.shape[4] * 2), mode='nearest')
            h = _conv(sd, q + '.conv', h, nm, trunk)
        elif (q + '.weight') in sd:        # the first input conv
            h = _conv(sd, q, h, nm, trunk)
        else:
            raise KeyError('cannot classify ' + q)
        if trace is not None:
            trace[q] = h
        j += 1
    return h


def _unet_trunk(sd, h, emb, context, heads, nm, trunk, trace=None, full3d=False):
    hs = []
    n_in = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('input_blocks.'))
    n_out = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('output_blocks.'))
    for i in range(n_in):
        h = _run_block(sd, f'input_blocks.{i}', h, emb, context, heads, nm, trunk, trace, full3d)
        hs.append(h)
    h = _run_block(sd, 'middle_block', h, emb, context, heads, nm, trunk, trace, full3d)
    for i in range(n_out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f'output_blocks.{i}', h, emb, context, heads, nm, trunk, trace, full3d)
    h = F.silu(F.group_norm(h, 32, sd['out.0.weight'], sd['out.0.bias'], 1e-5))
    return _conv(sd, 'out.2', h, nm, trunk)


# --------------------------------------------------------------------------------------
# a4/a5  UNet1DModel.forward (+ box_messsage_passing)   denoise_net.py:758-806
# --------------------------------------------------------------------------------------
def unet1d_forward(sd, box_t, obj_embed, triples, timesteps, heads=8, enable_t_emb=True, trace=None):
    """conditioning_key 'crossattn' (GCN output = the one cross-attention key) or 'concat' (GCN output appended to the
    box vector as input channels, denoise_net.py:789-790) -- told apart by the input conv's channel count."""
    mc = sd['time_embed.0.weight'].shape[1]
    t_emb = timestep_embedding(timesteps, mc)
    emb = _linear(sd, 'time_embed.2', F.silu(_linear(sd, 'time_embed.0', t_emb)))
    edges, p = _edges(triples)
    box_embed = _linear(sd, 'box_embeddings', box_t)
    pred_embed = sd['pred_embeddings.weight'][p]
    obj = torch.cat([obj_embed, box_embed], dim=1)
    if enable_t_emb:
        obj = torch.cat([obj, _linear(sd, 'box_time_emb', emb)], dim=1)
    ctx, _ = gcn_net(sd, 'box_graph_cov', obj, pred_embed, edges)
    context = ctx.unsqueeze(1)           # overwrites the caller's context (denoise_net.py:791-792)
    h = box_t.unsqueeze(1).permute(0, 2, 1)          # [O, 8, 1]
    if sd['input_blocks.0.0.weight'].shape[1] != box_t.shape[1]:        # 'concat'
        h = torch.cat([box_t, ctx], dim=1).unsqueeze(-1)
        context = None
    if trace is not None:
        trace.update(emb=emb, ctx=ctx, gcn_in=obj)
    out = _unet_trunk(sd, h, emb, context, heads, EXACT, False, trace)
    return out.squeeze(-1)


# --------------------------------------------------------------------------------------
# a12/a13  UNet3DModel.forward (+ shape_messsage_passing)   openai_model_3d.py:800-863
# --------------------------------------------------------------------------------------
def shape_stem(sd, x):
    """conv-pool stem: openai_model_3d.py:757-764 (MaxPool3d(2,2) then MaxPool3d(k=2, s=4))."""
    h = F.conv3d(x, sd['shape_embeddings.0.weight'], sd['shape_embeddings.0.bias'], padding=1)
    h = F.max_pool3d(h, kernel_size=2, stride=2)
    h = F.conv3d(h, sd['shape_embeddings.2.weight'], sd['shape_embeddings.2.bias'], padding=1)
    h = F.max_pool3d(h, kernel_size=2, stride=4)
    return _linear(sd, 'shape_embeddings.5', h.flatten(1))


def unet3d_forward(sd, x, obj_embed, triples, timesteps, context=None, heads=8,
                   enable_t_emb=True, nm=EXACT, trace=None, code_all=None, rows=None, c_concat=None):
    """sd: keys of UNet3DModel (i.e. without the 'diffusion_net.' prefix).

    ``code_all`` / ``rows`` (test support for the multi-GPU sharding, echoscene_amd/parallel.py): x holds only
    the objects ``rows`` of the graph; the conv-pool codes of ALL objects are supplied (all-gathered) so the
    GCN runs on the full graph, and the context rows of the local objects are kept."""
    mc = sd['time_embed.0.weight'].shape[1]
    t_emb = timestep_embedding(timesteps, mc)
    emb = _linear(sd, 'time_embed.2', F.silu(_linear(sd, 'time_embed.0', t_emb)))
    concat = c_concat is not None        # DiffusionUNet.forward 'concat' (diffusion_shape/network.py:26-28)
    if concat:
        x = torch.cat([x, c_concat.reshape(x.shape[0], -1, *x.shape[2:])], dim=1)
    if 'shape_code_graph_cov.gconvs.0.net1.0.weight' in sd:           # messsage_passing
        edges, p = _edges(triples)
        code = shape_stem(sd, x) if code_all is None else code_all
        obj = torch.cat([obj_embed.squeeze(1), code], dim=1)
        emb_full = emb
        if rows is not None:             # time embedding rows for every object of the graph (all share t)
            emb_full = emb[:1].expand(obj.shape[0], -1)
        if enable_t_emb:
            obj = torch.cat([obj, _linear(sd, 'shape_time_emb', emb_full)], dim=1)
        ctx, _ = gcn_net(sd, 'shape_code_graph_cov', obj, sd['pred_embeddings.weight'][p], edges)
        if rows is not None:
            ctx = ctx[rows]
        if concat:                       # GCN output becomes a fifth input channel (:841-842)
            x = torch.cat([x, ctx.view(-1, 1, *x.shape[2:])], dim=1)
            context = None
        else:
            context = ctx.unsqueeze(1)   # "we dont use the previous context" (:843-844)
        if trace is not None:
            trace.update(emb=emb, ctx=ctx, code=code)
    return _unet_trunk(sd, x, emb, context, heads, nm, True, trace, full3d=concat)


# --------------------------------------------------------------------------------------
# a9-a11  DDPM tables + ancestral sampling loop   diffusion_ddpm.py:38-84,133-162,220-345
# --------------------------------------------------------------------------------------
def get_betas(schedule_type, b_start, b_end, time_num):
    """get_betas (diffusion_ddpm.py:38-84).  'cosine' fails in the reference with UnboundLocalError (the table is computed but never
    bound, diffusion_ddpm.py:59-84): restated as that failure."""
    if schedule_type == 'linear':
        return np.linspace(b_start, b_end, time_num)
    if schedule_type in ('warm0.1', 'warm0.2', 'warm0.5'):
        frac = {'warm0.1': 0.1, 'warm0.2': 0.2, 'warm0.5': 0.5}[schedule_type]
        betas = b_end * np.ones(time_num, dtype=np.float64)
        warmup_time = int(time_num * frac)
        betas[:warmup_time] = np.linspace(b_start, b_end, warmup_time, dtype=np.float64)
        return betas
    if schedule_type == 'cosine':
        raise UnboundLocalError('betas')
    raise NotImplementedError(schedule_type)


def ddpm_tables(beta_start=1e-4, beta_end=0.02, time_num=1000, schedule_type='linear'):
    """Coefficient tables exactly as GaussianDiffusion.__init__ builds them: betas/cumprod in
    fp64 numpy, cast to fp32, *then* the derived tables in fp32 torch (the order of the casts
    matters for bit-parity)."""
    betas64 = get_betas(schedule_type, beta_start, beta_end, time_num).astype(np.float64)
    alphas64 = 1. - betas64
    ac = torch.from_numpy(np.cumprod(alphas64, axis=0)).float()
    ac_prev = torch.from_numpy(np.append(1., ac[:-1])).float()
    betas = torch.from_numpy(betas64).float()
    alphas = torch.from_numpy(alphas64).float()
    post_var = betas * (1. - ac_prev) / (1. - ac)
    return {
        'betas': betas,
        'posterior_variance': post_var,
        'sqrt_recip_alphas_cumprod': torch.sqrt(1. / ac).float(),
        'sqrt_recipm1_alphas_cumprod': torch.sqrt(1. / ac - 1).float(),
        'posterior_mean_coef1': betas * torch.sqrt(ac_prev) / (1. - ac),
        'posterior_mean_coef2': (1. - ac_prev) * torch.sqrt(alphas) / (1. - ac),
        'posterior_log_variance_clipped': torch.log(torch.max(post_var, 1e-20 * torch.ones_like(post_var))),
    }


def ddpm_step(tab, x, out, t, noise, clip_denoised=False, model_mean_type='eps', model_var_type='fixedsmall'):
    """p_mean_variance + p_sample_sg (diffusion_ddpm.py:220-264, 266-271, 204-217, 296-309).  ``out`` is the network's output: the
    noise ('eps', the shipped configs) or x_0 itself ('x0', :246-254); the log-variance is the clipped posterior's ('fixedsmall') or
    log(cat[posterior_variance[1:2], betas[1:]]) ('fixedlarge', :224-235); ``clip_denoised``: the predicted x0 clamped to [-1, 1]
    (:243-244; the shipped sampling call passes False)."""
    if model_mean_type == 'eps':
        x0 = tab['sqrt_recip_alphas_cumprod'][t] * x - tab['sqrt_recipm1_alphas_cumprod'][t] * out
    elif model_mean_type == 'x0':
        x0 = out
    else:
        raise NotImplementedError(model_mean_type)
    if clip_denoised:
        x0 = torch.clamp(x0, -1.0, 1.0)
    mean = tab['posterior_mean_coef1'][t] * x0 + tab['posterior_mean_coef2'][t] * x
    if model_var_type == 'fixedsmall':
        lv = tab['posterior_log_variance_clipped']
    elif model_var_type == 'fixedlarge':
        lv = torch.log(torch.cat([tab['posterior_variance'][1:2], tab['betas'][1:]]))
    else:
        raise NotImplementedError(model_var_type)
    logvar = lv[t] * torch.ones_like(x)
    nonzero = 1.0 - float(t == 0)
    return mean + nonzero * torch.exp(0.5 * logvar) * noise


def layout_sample_loop(sd, obj_embed, triples, noise, time_num=1000, n_steps=None,
                       beta_start=1e-4, beta_end=0.02, heads=8, enable_t_emb=True, trace=None, clip_denoised=False,
                       schedule_type='linear', model_mean_type='eps', model_var_type='fixedsmall'):
    """p_sample_loop_sg (diffusion_ddpm.py:330-345) with injected noise:
    noise[0] = x_T, noise[1+i] = draw of iteration i.  ``n_steps`` < time_num runs only the
    first n_steps iterations (t = time_num-1 ... time_num-n_steps) -- used for short goldens."""
    tab = ddpm_tables(beta_start, beta_end, time_num, schedule_type)
    O = obj_embed.shape[0]
    x = noise[0].clone()
    n_steps = time_num if n_steps is None else n_steps
    for i in range(n_steps):
        t = time_num - 1 - i
        t_ = torch.full((O,), t, dtype=torch.int64)
        out = unet1d_forward(sd, x, obj_embed, triples, t_, heads, enable_t_emb)
        x = ddpm_step(tab, x, out, t, noise[1 + i], clip_denoised, model_mean_type, model_var_type)
        if trace is not None:
            trace.append(x.clone())
    return x


# --------------------------------------------------------------------------------------
# a14-a16  shape schedule + DDIM   echo2shape.py:174-226, ldm_diffusion_util.py:43-96, ddim.py
# --------------------------------------------------------------------------------------
def shape_alphas_cumprod(linear_start=0.00085, linear_end=0.012, timesteps=1000):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    return torch.tensor(np.cumprod(1. - betas, axis=0), dtype=torch.float32)


def ddim_schedule(alphas_cumprod, S=100, ddpm_steps=1000):
    """make_ddim_timesteps('uniform') + make_ddim_sampling_parameters(eta=0)."""
    c = ddpm_steps // S
    ts = np.asarray(list(range(0, ddpm_steps, c))) + 1
    a = alphas_cumprod[ts]                                                 # fp32
    a_prev = np.asarray([alphas_cumprod[0].item()] + alphas_cumprod[ts[:-1]].tolist())
    return ts, a, torch.tensor(a_prev, dtype=torch.float32), torch.sqrt(1. - a)


def ddim_sigmas(alphas_cumprod, eta, S=100, ddpm_steps=1000):
    """sigma_t of make_ddim_sampling_parameters (ldm_diffusion_util.py:85-96): the published DDIM formula evaluated the way that
    function mixes its operands -- alphas a torch fp32 tensor, alphas_prev a float64 array -- and cast to fp32 by make_schedule."""
    ts, a, _, _ = ddim_schedule(alphas_cumprod, S, ddpm_steps)
    a_prev64 = np.asarray([alphas_cumprod[0].item()] + alphas_cumprod[ts[:-1]].tolist())
    # ``ndarray / Tensor`` is Tensor.__rtruediv__ = reciprocal(self) * other: 1 - alphas AND its reciprocal are formed in fp32,
    # everything after that in float64 (verified against the reference's tables bit for bit, tests/test_oracle_golden.py)
    recip = (1 - a).reciprocal().double().numpy()
    sig = eta * np.sqrt(recip * (1 - a_prev64) * (1 - a.double().numpy() / a_prev64))
    return torch.from_numpy(sig).to(torch.float32)


def ddim_step(x, e_t, a_t, a_prev, sqrt_1m_at, sigma_t=0.0, noise=None):
    """p_sample_ddim (ddim.py:236-262); sigma_t = 0 for eta = 0 (the shipped call), else + sigma_t * randn."""
    a_t = torch.as_tensor(a_t, dtype=torch.float32)
    a_prev = torch.as_tensor(a_prev, dtype=torch.float32)
    sigma_t = torch.as_tensor(sigma_t, dtype=torch.float32)
    pred_x0 = (x - sqrt_1m_at * e_t) / a_t.sqrt()
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma_t * noise
    return x_prev


def shape_sample_loop(sd, uc_s, triples, noise1, S=100, n_steps=None, heads=8, enable_t_emb=True,
                      nm=EXACT, linear_start=0.00085, linear_end=0.012, trace=None, c_concat=None, context=None, eta=0.0,
                      step_noise=None):
    """rel2shape's DDIM loop (echo2shape.py:484-521, ddim.py:127-181): one noise tensor shared by
    all objects, 'elif True' branch (single UNet call, no CFG); eta 0 unless given -- then ``step_noise`` [S, O, C,D,H,W] holds the
    per-step, per-object draws of p_sample_ddim (iteration order)."""
    ac = shape_alphas_cumprod(linear_start, linear_end)
    ts, a, a_prev, s1m = ddim_schedule(ac, S)
    sig = ddim_sigmas(ac, eta, S)
    O = uc_s.shape[0]
    x = noise1.repeat(O, 1, 1, 1, 1).clone()
    total = len(ts)
    n_steps = total if n_steps is None else n_steps
    for i in range(n_steps):
        index = total - 1 - i
        step = int(ts[index])
        t_ = torch.full((O,), step, dtype=torch.long)
        # ``context`` (c_s) only matters without message passing: the GCN output overwrites it otherwise
        e = unet3d_forward(sd, x, uc_s, triples, t_, context, heads, enable_t_emb, nm, c_concat=c_concat)
        x = ddim_step(x, e, a[index], a_prev[index], s1m[index], sig[index], None if eta == 0.0 else step_noise[i])
        if trace is not None:
            trace.append(x.clone())
    return x


# --------------------------------------------------------------------------------------
# a18  VQ-VAE decode_no_quant   vqvae_networks/network.py:95-103, quantizer.py:68-119,
#      vqvae_modules.py:67-126 (ResnetBlock), 128-176 (AttnBlock), 376-409 (Decoder3D.forward)
# --------------------------------------------------------------------------------------
def _vq_norm(sd, p, x):
    C = x.shape[1]
    g = C // 4 if C <= 32 else (32 if C % 32 == 0 else 30)          # vqvae_modules.py:13-21
    return F.group_norm(x, g, sd[p + '.weight'], sd[p + '.bias'], 1e-6)


def _vq_res(sd, p, x):
    h = F.conv3d(F.silu(_vq_norm(sd, p + '.norm1', x)), sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], padding=1)
    h = F.conv3d(F.silu(_vq_norm(sd, p + '.norm2', h)), sd[p + '.conv2.weight'], sd[p + '.conv2.bias'], padding=1)
    if (p + '.nin_shortcut.weight') in sd:
        x = F.conv3d(x, sd[p + '.nin_shortcut.weight'], sd[p + '.nin_shortcut.bias'])
    return x + h


def _vq_attn(sd, p, x):
    h = _vq_norm(sd, p + '.norm', x)
    q = F.conv3d(h, sd[p + '.q.weight'], sd[p + '.q.bias'])
    k = F.conv3d(h, sd[p + '.k.weight'], sd[p + '.k.bias'])
    v = F.conv3d(h, sd[p + '.v.weight'], sd[p + '.v.bias'])
    B, C = q.shape[:2]
    q, k, v = q.reshape(B, C, -1), k.reshape(B, C, -1), v.reshape(B, C, -1)
    w = torch.bmm(q.permute(0, 2, 1), k) * (int(C) ** -0.5)
    w = F.softmax(w, dim=2)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(x.shape)
    return x + F.conv3d(h, sd[p + '.proj_out.weight'], sd[p + '.proj_out.bias'])


def vq_quantize(sd, z):
    """Nearest-codebook lookup; distance expression as in quantizer.py:80-84."""
    B, C = z.shape[:2]
    zf = z.permute(0, 2, 3, 4, 1).contiguous().view(-1, C)
    E = sd['quantize.embedding.weight']
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(E ** 2, dim=1) - 2 * (zf @ E.t())
    idx = torch.argmin(d, dim=1)
    zq = E[idx].view(B, *z.shape[2:], C).permute(0, 4, 1, 2, 3).contiguous()
    return zq, idx


def vqvae_decode_no_quant(sd, z):
    zq, _ = vq_quantize(sd, z)
    h = F.conv3d(zq, sd['post_quant_conv.weight'], sd['post_quant_conv.bias'])
    d = _sub(sd, 'decoder.')
    h = F.conv3d(h, d['conv_in.weight'], d['conv_in.bias'], padding=1)
    h = _vq_res(d, 'mid.block_1', h)
    h = _vq_attn(d, 'mid.attn_1', h)
    h = _vq_res(d, 'mid.block_2', h)
    n_lvl = 1 + max(int(k.split('.')[1]) for k in d if k.startswith('up.'))
    for lvl in reversed(range(n_lvl)):
        b = 0
        while f'up.{lvl}.block.{b}.norm1.weight' in d:
            h = _vq_res(d, f'up.{lvl}.block.{b}', h)
            b += 1
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = F.conv3d(h, d[f'up.{lvl}.upsample.conv.weight'], d[f'up.{lvl}.upsample.conv.bias'], padding=1)
    h = F.gelu(_vq_norm(d, 'norm_out', h))
    return F.conv3d(h, d['conv_out.weight'], d['conv_out.bias'], padding=1)


# --------------------------------------------------------------------------------------
# a19  one-off setup: init_encoder, manipulate, rel_s_mlp   model/EchoScene.py:143-195,388-420
#      (EchoLayout differs only in the manipulator's predicate table, EchoLayout.py:154)
# --------------------------------------------------------------------------------------
def scene_setup(sd, objs, triples, text_feat, rel_feat, model_type='echoscene', change_noise=None,
                embedding_dim=64, clip=True):
    """Returns obj_embed_ (uc_b) and latent (c_b) for a plain ``sample`` (no edits; with ``change_noise`` [O, 64] the manipulator
    output of sample_with_changes, EchoScene.py:422-439).  ``clip=False``: no CLIP features, the node / predicate vectors are the
    embeddings alone (EchoScene.py:151-153,188-190)."""
    edges, p = _edges(triples)
    cat = (lambda f, e: torch.cat([f, e], dim=1)) if clip else (lambda f, e: e)
    obj_embed = cat(text_feat, sd['obj_embeddings_ec.weight'][objs])
    pred_embed = cat(rel_feat, sd['pred_embeddings_ec.weight'][p])
    latent, _ = gcn_net(sd, 'gconv_net_ec', obj_embed, pred_embed, edges)
    change = torch.zeros(objs.shape[0], embedding_dim) if change_noise is None else change_noise
    latent_ = torch.cat([latent, change], dim=1)
    ptab = 'pred_embeddings_ec.weight' if model_type == 'echoscene' else 'pred_embeddings_man_dc.weight'
    pred_embed_m = cat(rel_feat, sd[ptab][p])
    man_in = torch.cat([latent_, obj_embed], dim=1)
    latent_m, _ = gcn_net(sd, 'gconv_net_manipulation', man_in, pred_embed_m, edges)
    return obj_embed, latent_m, latent


def rel_s(sd, feat):
    """rel_s_mlp: Linear-BN-ReLU-Linear (norelu) then unsqueeze(1)  (EchoScene.py:97-100,413-416)."""
    return mlp(sd, 'rel_s_mlp', feat, final_nonlinearity=False).unsqueeze(1)


# --------------------------------------------------------------------------------------
# (f3) post-path box helpers   helpers/util.py:542-557 (descale_box_params), :559-568
# --------------------------------------------------------------------------------------
def descale_box_params(boxes, stats, angle=False):
    stats = torch.as_tensor(stats, dtype=boxes.dtype)
    min_lhw, max_lhw, min_xyz, max_xyz, min_angle, max_angle = stats[:3], stats[3:6], stats[6:9], stats[9:12], stats[12:13], stats[13:]
    out = boxes.clone()
    out[:, :3] = (out[:, :3] + 1) / 2
    out[:, :3] = out[:, :3] * (max_lhw - min_lhw) + min_lhw
    out[:, 3:6] = (out[:, 3:6] + 1) / 2
    out[:, 3:6] = out[:, 3:6] * (max_xyz - min_xyz) + min_xyz
    if angle:                                                       # helpers/util.py:553-555
        out[:, 6:7] = (out[:, 6:7] + 1) / 2
        out[:, 6:7] = out[:, 6:7] * (max_angle - min_angle) + min_angle
    return out


def sincos2arctan(sincos):
    return torch.arctan2(sincos[:, 0], sincos[:, 1]).reshape(-1, 1)


# --------------------------------------------------------------------------------------
# (f4) chamfer nearest-neighbour distance   extension/old_chamfer/chamfer.cu:12-134 (forward),
#      :155-174 (backward): squared distance to the nearest point of the other cloud + its index
# --------------------------------------------------------------------------------------
def chamfer_forward(xyz1, xyz2):
    d = ((xyz1[:, :, None, :] - xyz2[:, None, :, :]) ** 2)
    d = d[..., 0] + d[..., 1] + d[..., 2]                    # dx*dx + dy*dy + dz*dz, fp32
    dist1, idx1 = d.min(dim=2)
    dist2, idx2 = d.min(dim=1)
    return dist1, idx1, dist2, idx2


def chamfer_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    gx1, gx2 = torch.zeros_like(xyz1), torch.zeros_like(xyz2)
    for b in range(xyz1.shape[0]):
        v = 2 * g1[b][:, None] * (xyz1[b] - xyz2[b][idx1[b]])
        gx1[b] += v
        gx2[b].index_add_(0, idx1[b], -v)
        v = 2 * g2[b][:, None] * (xyz2[b] - xyz1[b][idx2[b]])
        gx2[b] += v
        gx1[b].index_add_(0, idx2[b], -v)
    return gx1, gx2
