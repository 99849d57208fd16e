"""GPU parity tests of the "rows" path (GCN + layout denoiser), all through the C ABI.

 * unit level: es_linear_rows_f32 against a plain PyTorch fp32 CPU evaluation of the same fused op
   (tolerance 2e-5 * scale: same fp32 arithmetic, different summation order);
 * network level: HIP vs the reference-generated golden vectors and vs the CPU oracle.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, seeded_state_dict
from echoscene_amd import synth, config as escfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda')


def _close(a, b, atol, rtol=1e-5):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert torch.isfinite(a).all()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), 'max abs err %.3e (ref scale %.3e)' % (err, b.abs().max().item())


def _run_linear(dev, segs_cpu, W, b, M, prologue=0, gamma=None, beta=None, eps=0.0, act=0, res=None, res2=None):
    """segs_cpu: list of dicts(src=tensor, mode, idx, ent_row, ent_off, width, col)."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, View, seg
    bld = Builder(dev)
    pl = PackedLinear(W, b, dev)
    segs = []
    for s in segs_cpu:
        t = bld.dev(s['src'])
        v = View(t, col=s.get('col', 0), ld=s.get('ld', None), width=s['width'])
        mk = lambda a: None if a is None else bld.dev(a, torch.int32)
        segs.append(seg(v, s.get('mode', 0), idx=mk(s.get('idx')), ent_row=mk(s.get('ent_row')),
                        ent_off=mk(s.get('ent_off')), ent_wt=None if s.get('ent_wt') is None else View(bld.dev(s['ent_wt']))))
    out = bld.buf(M, pl.N, zero=True)
    g = None if gamma is None else bld.dev(gamma)
    be = None if beta is None else bld.dev(beta)
    r1 = None if res is None else View(bld.dev(res))
    r2 = None if res2 is None else View(bld.dev(res2))
    bld.linear(segs, pl, M, View(out), prologue=prologue, gamma=g, beta=be, eps=eps, act=act, res=r1, res2=r2)
    plan = bld.finish()
    plan.run()
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize('M,K,N', [(5, 8, 8), (32, 512, 512), (32, 2048, 96), (124, 1664, 256), (33, 640, 24)])
def test_linear_plain(dev, M, K, N):
    rs = np.random.RandomState(M * 7 + N)
    X = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32))
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32))
    R = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32))
    out = _run_linear(dev, [dict(src=X, width=K)], W, b, M, act=1, res=R)
    _close(out, F.relu(F.linear(X, W, b)) + R, 2e-5)


def test_linear_prologues(dev):
    from echoscene_amd import hip
    rs = np.random.RandomState(3)
    M, K, N = 32, 512, 64
    X = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)) * 1.7 + 0.3
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32))
    ga = torch.from_numpy(1 + 0.1 * rs.standard_normal(K).astype(np.float32))
    be = torch.from_numpy(0.1 * rs.standard_normal(K).astype(np.float32))
    gn = lambda eps: F.group_norm(X.unsqueeze(-1), 32, ga, be, eps).squeeze(-1)
    _close(_run_linear(dev, [dict(src=X, width=K)], W, b, M, hip.PRO_GN_SILU, ga, be, 1e-5),
           F.linear(F.silu(gn(1e-5)), W, b), 2e-5)
    _close(_run_linear(dev, [dict(src=X, width=K)], W, b, M, hip.PRO_GN, ga, be, 1e-6, act=hip.ACT_SILU),
           F.silu(F.linear(gn(1e-6), W, b)), 2e-5)
    _close(_run_linear(dev, [dict(src=X, width=K)], W, b, M, hip.PRO_LN, ga, be, 1e-5),
           F.linear(F.layer_norm(X, (K,), ga, be, 1e-5), W, b), 2e-5)
    _close(_run_linear(dev, [dict(src=X, width=K)], W, b, M, hip.PRO_SILU), F.linear(F.silu(X), W, b), 2e-5)
    # GEGLU: source holds [value | gate]
    X2 = torch.from_numpy(rs.standard_normal((M, 2 * K)).astype(np.float32))
    a, g = X2.chunk(2, dim=-1)
    R2 = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32))
    _close(_run_linear(dev, [dict(src=X2, width=K, ld=2 * K)], W, b, M, hip.PRO_GEGLU, res=R2, res2=R2),
           F.linear(a * F.gelu(g), W, b) + 2 * R2, 2e-5)
    # concat of two sources under one GroupNorm (skip connections of the output blocks)
    Xa, Xb = X[:, :256].contiguous(), X[:, 256:].contiguous()
    _close(_run_linear(dev, [dict(src=Xa, width=256), dict(src=Xb, width=256)], W, b, M, hip.PRO_GN_SILU, ga, be, 1e-5),
           F.linear(F.silu(gn(1e-5)), W, b), 2e-5)


def test_linear_gather_and_csr_mean(dev):
    from echoscene_amd import hip
    rs = np.random.RandomState(5)
    O, T, D, Dp, H = 9, 31, 48, 16, 32
    obj = torch.from_numpy(rs.standard_normal((O, D)).astype(np.float32))
    pred = torch.from_numpy(rs.standard_normal((T, Dp)).astype(np.float32))
    s = torch.from_numpy(rs.randint(0, O - 1, T)).long()          # node O-1 appears in no triple
    o = torch.from_numpy(rs.randint(0, O - 1, T)).long()
    K = 2 * D + Dp
    W = torch.from_numpy((rs.standard_normal((H, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.zeros(H)
    out = _run_linear(dev, [dict(src=obj, width=D, mode=hip.SEG_GATHER, idx=s), dict(src=pred, width=Dp),
                            dict(src=obj, width=D, mode=hip.SEG_GATHER, idx=o)], W, b, T)
    _close(out, F.linear(torch.cat([obj[s], pred, obj[o]], 1), W, b), 2e-5)
    # CSR mean == scatter_add/avg of graph.py:172-199
    msg = torch.from_numpy(rs.standard_normal((T, 2 * H + Dp)).astype(np.float32))
    pooled = torch.zeros(O, H).index_add_(0, s, msg[:, :H]).index_add_(0, o, msg[:, H + Dp:])
    cnt = torch.zeros(O).index_add_(0, s, torch.ones(T)).index_add_(0, o, torch.ones(T)).clamp(min=1)
    pooled = pooled / cnt[:, None]
    rows, offs, ptr = [], [], [0]
    for n in range(O):
        ts, to = (s == n).nonzero().flatten().tolist(), (o == n).nonzero().flatten().tolist()
        rows += ts + to
        offs += [0] * len(ts) + [H + Dp] * len(to)
        ptr.append(len(rows))
    W2 = torch.from_numpy((rs.standard_normal((24, H)) / np.sqrt(H)).astype(np.float32))
    out = _run_linear(dev, [dict(src=msg, width=H, ld=2 * H + Dp, mode=hip.SEG_CSRMEAN, idx=torch.tensor(ptr),
                                 ent_row=torch.tensor(rows), ent_off=torch.tensor(offs))], W2, None, O)
    _close(out, F.linear(pooled, W2), 2e-5)
    assert out[O - 1].abs().max() == 0          # isolated node pools to exactly zero
    # pooling='sum' (graph.py:105,186: the scatter_add result without the division)
    out = _run_linear(dev, [dict(src=msg, width=H, ld=2 * H + Dp, mode=hip.SEG_CSRSUM, idx=torch.tensor(ptr),
                                 ent_row=torch.tensor(rows), ent_off=torch.tensor(offs))], W2, None, O)
    _close(out, F.linear(pooled * cnt[:, None], W2), 2e-5)


@pytest.mark.parametrize('N,K,geglu', [(512, 512, False), (8, 512, False), (2, 128, False), (100, 72, False), (4096, 512, True), (48, 40, True)])
def test_weight_relayout_on_the_device_equals_the_host_loop(dev, N, K, geglu):
    """es_pack_linear_f32_dev (what the planner uses) against es_pack_linear_f32 / es_pack_linear_geglu_f32: identical images."""
    from echoscene_amd.plan import PackedLinear
    rs = np.random.RandomState(N + K)
    W = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32))
    d, h = PackedLinear(W, b, dev, geglu=geglu), PackedLinear(W, b, 'cpu', geglu=geglu)
    torch.cuda.synchronize()
    assert d.w.is_cuda and torch.equal(d.w.cpu(), h.w) and torch.equal(d.b.cpu(), h.b)


def test_bad_args_raise(dev):
    from echoscene_amd import hip
    X = torch.zeros(4, 6)
    with pytest.raises(RuntimeError, match='multiples of 4'):
        _run_linear(dev, [dict(src=X, width=6)], torch.zeros(8, 6), None, 4)


@pytest.mark.parametrize('tag', ['res_bn', 'plain', 'ragged', 'one_node'])
def test_gcn_vs_reference_golden(dev, tag):
    from echoscene_amd.model.graph import GraphTripleConvNet
    from echoscene_amd.samplers import gcn_forward
    g = load_golden('gcn_' + tag)
    din, dp, nl, H, res, bn, dout = [int(v) for v in g['cfg']]
    net = GraphTripleConvNet(din, dp, num_layers=nl, hidden_dim=H, residual=bool(res),
                             mlp_normalization='batch' if bn else 'none', output_dim=dout)
    # ('ragged': a node without triples -- avg pooling over an empty CSR row --, a hub, a repeated pair, a self-loop; 'one_node': one
    #  node with one self-loop triple; both with the weights of 'res_bn')
    sd = {'n.' + k: v for k, v in seeded_state_dict(net, 'gcn_%s.' % (tag if tag in ('res_bn', 'plain') else 'res_bn')).items()}
    o, p = gcn_forward(sd, 'n', g['obj'], g['pred'], g['triples'], dev)
    _close(o, g['out_obj'], 3e-5)
    _close(p, g['out_pred'], 3e-5)


@pytest.mark.parametrize('tag', ['sum_g8', 'sum_ragged', 'wavg_g8', 'wavg_ragged'])
def test_gcn_other_poolings_vs_reference_golden(dev, tag):
    """pooling='sum' and pooling='wAvg' (the learned weights of WeightNetGCN, model/graph.py:37-86, 163-184: SEG_CSRWAVG +
    ACT_SIGMOID) against the reference's GraphTripleConvNet."""
    from echoscene_amd.model.graph import GraphTripleConvNet
    from echoscene_amd.samplers import gcn_forward
    g = load_golden('gcn_' + tag)
    din, dp, nl, H, res, bn, dout, code = [int(v) for v in g['cfg']]
    pool = ('avg', 'sum', 'wAvg')[code]
    net = GraphTripleConvNet(din, dp, num_layers=nl, hidden_dim=H, residual=bool(res), pooling=pool,
                             mlp_normalization='batch' if bn else 'none', output_dim=dout)
    sd = {'n.' + k: v for k, v in seeded_state_dict(net, 'gcn_%s.' % pool).items()}
    o, p = gcn_forward(sd, 'n', g['obj'], g['pred'], g['triples'], dev, pooling=pool)
    _close(o, g['out_obj'], 3e-5)
    _close(p, g['out_pred'], 3e-5)


def test_csr_weighted_mean(dev):
    """SEG_CSRWAVG alone: sum_e w_e v_e / (sum_e w_e + 1e-4) with the reference's rounding points (the product w * v is rounded,
    then summed in scatter_add order; the weights are summed object slots first), model/graph.py:169-184."""
    from echoscene_amd import hip
    rs = np.random.RandomState(9)
    O, T, Dp, H = 7, 23, 8, 32
    s = torch.from_numpy(rs.randint(0, O - 1, T)).long()          # node O-1 appears in no triple
    o = torch.from_numpy(rs.randint(0, O - 1, T)).long()
    msg = torch.from_numpy(rs.standard_normal((T, 2 * H + Dp)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(0.05, 0.95, (T, 2)).astype(np.float32))
    pooled = torch.zeros(O, H).index_add_(0, s, w[:, :1] * msg[:, :H]).index_add_(0, o, w[:, 1:] * msg[:, H + Dp:])
    wsum = torch.zeros(O, 1).index_add_(0, o, w[:, 1:]).index_add_(0, s, w[:, :1])
    pooled = pooled / (wsum + 0.0001)
    rows, offs, ptr = [], [], [0]
    for n in range(O):
        ts, to = (s == n).nonzero().flatten().tolist(), (o == n).nonzero().flatten().tolist()
        rows += ts + to
        offs += [0] * len(ts) + [H + Dp] * len(to)
        ptr.append(len(rows))
    W2 = torch.eye(H)                                             # the product with the identity is exact: the pooled rows themselves
    out = _run_linear(dev, [dict(src=msg, width=H, ld=2 * H + Dp, mode=hip.SEG_CSRWAVG, idx=torch.tensor(ptr),
                                 ent_row=torch.tensor(rows), ent_off=torch.tensor(offs), ent_wt=w)], W2, None, O)
    assert torch.equal(out.cpu(), pooled), (out.cpu() - pooled).abs().max()
    assert out[O - 1].abs().max() == 0


def _layout(dev, mc, ctx, prefix, time_num, t_emb=True):
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    kw = dict(escfg.layout_denoiser_kwargs(mc))
    kw['concat_dim'] = kw['crossattn_dim'] = ctx
    if not t_emb:
        del kw['enable_t_emb']              # config/box.yaml, config/full.yaml: no key -> the constructor's default (False)
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix=prefix)
    return LayoutDenoiser(net, escfg.layout_diffusion_kwargs(time_num), dev)


def test_unet1d_tiny_eps_vs_reference_golden(dev):
    g = load_golden('unet1d_tiny')
    den = _layout(dev, 128, 128, 'unet1d_tiny.', 1000)
    t = int(g['t'][0])
    eps = den.eps(g['box'], g['obj_embed'], g['triples'], iteration=999 - t)
    _close(eps, g['eps'], 5e-5)


@pytest.mark.parametrize('use_graph', [False, True])
def test_layout_loop_tiny_100_steps_vs_reference_golden(dev, use_graph):
    """BASELINE.json configs[0] on the HIP path: 8 nodes, 100 DDPM steps, injected noise."""
    g = load_golden('layout_loop_tiny')
    den = _layout(dev, 128, 128, 'unet1d_tiny.', 100)
    noise = synth.layout_noise(8, 8, 100, seed=7)
    x = den.sample(g['obj_embed'], g['triples'], noise, use_graph=use_graph)
    _close(x, g['x_final'], 2e-4)
    x2 = den.sample(g['obj_embed'], g['triples'], noise, use_graph=use_graph)
    assert torch.equal(x, x2), 'sampling must be deterministic for fixed noise'
    # the plan's scratch claim (model files store such buffers empty): poisoned with NaN patterns, the loop gives the same bits
    den._last['plan'].poison_scratch()                 # (the layout planner marks no buffer yet: the call is the contract)
    x3 = den.sample(g['obj_embed'], g['triples'], noise, use_graph=use_graph)
    assert torch.equal(x, x3), 'an op reads scratch bytes that no op of the plan wrote'


@pytest.mark.parametrize('shape', [(672, 672, 2688), (100, 37, 70), (5, 3, 1), (1, 1000, 1)])
def test_fold_product_on_the_device(dev, shape):
    """plan.mm64 on device tensors (es_matmul_f64: a fixed left fold over k per element) against the host fp64 product, a vector
    right-hand side included; two calls leave the same bits."""
    from echoscene_amd.plan import mm64
    N, K, M = shape
    rs = np.random.RandomState(N + K + M)
    A = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32))
    B = torch.from_numpy(rs.standard_normal((K, M)).astype(np.float32))
    ref = A.double() @ B.double()
    got = mm64(A.to(dev), B.to(dev))
    got2 = mm64(A.to(dev), B.to(dev))
    torch.cuda.synchronize()
    assert got.dtype == torch.float64 and got.is_cuda and torch.equal(got, got2)
    assert float((got.cpu() - ref).abs().max()) <= 1e-12 * max(1.0, float(ref.abs().max()))
    if M == 1:
        v = mm64(A.to(dev), B[:, 0].to(dev))
        assert v.shape == (N,) and torch.equal(v, got[:, 0])


def test_layout_denoiser_from_a_model_on_the_gpu(dev):
    """The planners read a model's parameters IN PLACE when it already sits on the GPU (samplers.state_dict_for: fp64 folds and
    re-layouts on the device, no download / upload of every tensor): same golden, and the plan owns everything it keeps -- zeroing
    the module's parameters afterwards changes nothing."""
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    g = load_golden('unet1d_full')
    kw = dict(escfg.layout_denoiser_kwargs(512))
    kw['concat_dim'] = kw['crossattn_dim'] = 1280
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='unet1d_full.')
    net.to(dev)
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    eps = den.eps(g['box8'], g['obj_embed8'], g['triples8'], iteration=999 - 617)
    _close(eps, g['eps8'], 1e-4)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.zero_()
    torch.cuda.synchronize()
    eps2 = den.eps(g['box8'], g['obj_embed8'], g['triples8'], iteration=999 - 617)
    assert torch.equal(eps, eps2)


def test_unet1d_full_vs_reference_golden(dev):
    """Full-width layout denoiser (config/full_mp.yaml) at O=8 and O=32 (BASELINE configs[1] size)."""
    g = load_golden('unet1d_full')
    den = _layout(dev, 512, 1280, 'unet1d_full.', 1000)
    for O in (8, 32):
        eps = den.eps(g['box%d' % O], g['obj_embed%d' % O], g['triples%d' % O], iteration=999 - 617)
        _close(eps, g['eps%d' % O], 1e-4)
    noise = synth.layout_noise(8, 8, 1000, seed=7)[:11]
    x = den.sample(g['loop_obj_embed'], g['loop_triples'], noise, n_steps=10)
    _close(x, g['loop_x10'], 2e-4)


def test_layout_denoiser_without_time_embedding_vs_reference_golden(dev):
    """config/box.yaml / config/full.yaml build the layout denoiser without ``enable_t_emb`` (denoise_net.py:505,735-740,766-768):
    the box GCN's node vectors are [obj_embed 640 | box 64], no box_time_emb.  Goldens from the reference module: the whole 100-step
    tiny loop; full width: eps at O = 8 / O = 32 (BASELINE configs[1] size) and 10 steps of the 1000-step loop."""
    g = load_golden('layout_loop_tiny_no_temb')
    den = _layout(dev, 128, 128, 'unet1d_tiny_no_temb.', 100, t_emb=False)
    assert not den.net.enable_t_emb and den.w.box_t is None
    x = den.sample(g['obj_embed'], g['triples'], synth.layout_noise(8, 8, 100, seed=7))
    _close(x, g['x_final'], 2e-4)
    g = load_golden('unet1d_full_no_temb')
    den = _layout(dev, 512, 1280, 'unet1d_full_no_temb.', 1000, t_emb=False)
    for O in (8, 32):
        eps = den.eps(g['box%d' % O], g['obj_embed%d' % O], g['triples%d' % O], iteration=999 - 617)
        _close(eps, g['eps%d' % O], 1e-4)
    x = den.sample(g['loop_obj_embed'], g['loop_triples'], synth.layout_noise(8, 8, 1000, seed=7)[:11], n_steps=10)
    _close(x, g['loop_x10'], 2e-4)


def test_layout_loop_with_clip_denoised_vs_reference_golden(dev):
    """clip_denoised=True (gen_samples_sg -> p_mean_variance clamps the predicted x0 to [-1, 1], diffusion_ddpm.py:243-244): all 100
    steps at tiny width against the reference's own loop (make_golden.py case_sampler_options); through LayoutDenoiser and through the
    drop-in EchoToLayout.generate_layout_sg, which also accepts ret_traj / ddim and -- like the reference -- ignores them."""
    g = load_golden('sampler_options_tiny')
    den = _layout(dev, 128, 128, 'unet1d_tiny.', 100)
    noise = synth.layout_noise(8, 8, 100, seed=7)
    x = den.sample(g['layout_obj_embed'], g['layout_triples'], noise, clip_denoised=True)
    _close(x, g['layout_x_final_clip'], 2e-4)
    x_plain = den.sample(g['layout_obj_embed'], g['layout_triples'], noise)
    _close(x_plain, load_golden('layout_loop_tiny')['x_final'], 2e-4)      # the unclipped plan is a separate cache entry


def test_layout_loop_sampler_variants_vs_reference_golden(dev):
    """The warm-up beta schedules, x0-prediction and the 'fixedlarge' variance of GaussianDiffusion (diffusion_ddpm.py:38-58, 224-235,
    246-254) on the HIP path: all 100 steps at tiny width against the reference's own gen_samples_sg (make_golden.py
    case_sampler_variants).  Every variant is a coefficient table of the same update kernel (schedules.LayoutSchedule)."""
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    g = load_golden('sampler_variants_tiny')
    kw = dict(escfg.layout_denoiser_kwargs(128))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='unet1d_tiny.')
    noise = synth.layout_noise(8, 8, 100, seed=7)
    for tag, ov, clip in (('warm01_large', dict(schedule_type='warm0.1', model_var_type='fixedlarge'), False),
                          ('warm05_x0', dict(schedule_type='warm0.5', model_mean_type='x0'), False),
                          ('warm02_x0_large_clip', dict(schedule_type='warm0.2', model_mean_type='x0', model_var_type='fixedlarge'), True)):
        dk = dict(escfg.layout_diffusion_kwargs(100))
        dk.update(ov)
        den = LayoutDenoiser(net, dk, dev)
        x = den.sample(g['obj_embed'], g['triples'], noise, clip_denoised=clip)
        _close(x, g['x_final_' + tag], 2e-4)


def test_unet1d_tiny_blockwise_vs_oracle(dev):
    """Every intermediate of the tiny layout denoiser (time MLP, GCN context, each ResBlock /
    transformer / resample output) against the CPU oracle -- localises any mismatch."""
    from oracle import echoscene_oracle as orc
    g = load_golden('unet1d_tiny')
    den = _layout(dev, 128, 128, 'unet1d_tiny.', 1000)
    t = int(g['t'][0])
    den.eps(g['box'], g['obj_embed'], g['triples'], iteration=999 - t)
    st = next(iter(den._plans.values()))
    trace = {}
    sd = {k: v.detach().cpu() for k, v in den.net.state_dict().items()}
    orc.unet1d_forward(sd, g['box'], g['obj_embed'], g['triples'], g['t'], trace=trace)
    bad = []
    for name, v in st['eps_plan'].tags.items():
        if name == 'eps':
            continue                        # (the step's result: checked by the golden tests)
        ref = trace[name]
        ref = ref.reshape(ref.shape[0], -1)
        got = v.value().cpu()                     # (slab tensors: the fixed-order sum of their slabs)
        err = (got - ref).abs().max().item()
        if not err < 1e-4 * max(1.0, ref.abs().max().item()):
            bad.append((name, '%.2e' % err, 'rows', (got - ref).abs().max(1).values.gt(1e-4).nonzero().flatten().tolist()))
    assert not bad, bad[:6]


@pytest.mark.parametrize('K,pro', [(128, 'gn'), (256, 'gn'), (64, 'ln'), (192, 'ln'), (1024, 'gn'), (1024, 'ln')])
def test_linear_narrow_norms(dev, K, pro):
    from echoscene_amd import hip
    rs = np.random.RandomState(K)
    M, N = 8, 40
    X = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)) * 2 - 0.5
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    ga = torch.from_numpy(1 + 0.1 * rs.standard_normal(K).astype(np.float32))
    be = torch.from_numpy(0.1 * rs.standard_normal(K).astype(np.float32))
    if pro == 'gn':
        ref = F.linear(F.silu(F.group_norm(X.unsqueeze(-1), 32, ga, be, 1e-5).squeeze(-1)), W)
        out = _run_linear(dev, [dict(src=X, width=K)], W, None, M, hip.PRO_GN_SILU, ga, be, 1e-5)
    else:
        ref = F.linear(F.layer_norm(X, (K,), ga, be, 1e-5), W)
        out = _run_linear(dev, [dict(src=X, width=K)], W, None, M, hip.PRO_LN, ga, be, 1e-5)
    _close(out, ref, 2e-5)


def test_linear_multisegment_multichunk(dev):
    """gather | direct | gather with K > 1024 (the GCN's first layer at full width: K = 1664)."""
    from echoscene_amd import hip
    rs = np.random.RandomState(9)
    O, T, D, Dp, H = 8, 29, 768, 128, 256
    obj = torch.from_numpy(rs.standard_normal((O, D)).astype(np.float32))
    pred = torch.from_numpy(rs.standard_normal((T, Dp)).astype(np.float32))
    s = torch.from_numpy(rs.randint(0, O, T)).long()
    o = torch.from_numpy(rs.randint(0, O, T)).long()
    K = 2 * D + Dp
    W = torch.from_numpy((rs.standard_normal((H, K)) / np.sqrt(K)).astype(np.float32))
    out = _run_linear(dev, [dict(src=obj, width=D, mode=hip.SEG_GATHER, idx=s), dict(src=pred, width=Dp),
                            dict(src=obj, width=D, mode=hip.SEG_GATHER, idx=o)], W, None, T, act=hip.ACT_RELU)
    _close(out, F.relu(F.linear(torch.cat([obj[s], pred, obj[o]], 1), W)), 2e-5)


def test_linear_geglu_epilogue_and_batched(dev):
    """GEGLU fused into the projection's epilogue (interleaved weight tiles) and the batched launch used for the
    11 cross-attention output projections."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, PackedLinearBatch, View, seg
    rs = np.random.RandomState(11)
    M, K, Nh = 32, 512, 2048
    X = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32))
    W = torch.from_numpy((rs.standard_normal((2 * Nh, K)) / np.sqrt(K)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(2 * Nh).astype(np.float32))
    ga = torch.from_numpy(1 + 0.1 * rs.standard_normal(K).astype(np.float32))
    be = torch.from_numpy(0.1 * rs.standard_normal(K).astype(np.float32))
    h = F.linear(F.layer_norm(X, (K,), ga, be, 1e-5), W, bias)
    a, g = h.chunk(2, -1)
    b = Builder(dev)
    out = b.buf(M, Nh, zero=True)
    b.linear([seg(View(b.dev(X)))], PackedLinear(W, bias, dev, geglu=True), M, View(out), prologue=hip.PRO_LN,
             gamma=b.dev(ga), beta=b.dev(be), eps=1e-5)
    nb, C = 5, 64
    A = torch.from_numpy(rs.standard_normal((M, nb * C)).astype(np.float32))
    Ws = [torch.from_numpy((rs.standard_normal((C, C)) / 8).astype(np.float32)) for _ in range(nb)]
    bs = [torch.from_numpy(rs.standard_normal(C).astype(np.float32)) for _ in range(nb)]
    outb = b.buf(nb, M, C, zero=True)
    b.linear([seg(View(b.dev(A), ld=nb * C, width=C))], PackedLinearBatch(Ws, bs, dev), M, View(outb.view(nb * M, C)),
             a_bstride=C, out_bstride=M * C)
    b.finish().run()
    torch.cuda.synchronize()
    _close(out, a * F.gelu(g), 3e-5)
    for z in range(nb):
        _close(outb[z], F.linear(A[:, z * C:(z + 1) * C], Ws[z], bs[z]), 2e-5)


# ---- 'concat'-conditioned layout denoiser (config/full_concat_mp.yaml) ----
@pytest.mark.parametrize('tag,mc,cd', [('tiny', 128, 128), ('full', 512, 1280)])
def test_unet1d_concat_vs_reference_golden(dev, tag, mc, cd):
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    g = load_golden('unet1d_concat_' + tag)
    kw = dict(escfg.layout_denoiser_kwargs(mc, concat=True))
    kw['concat_dim'] = kw['crossattn_dim'] = cd
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='unet1d_concat_%s.' % tag)
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    eps = den.eps(g['box'], g['obj_embed'], g['triples'], iteration=999 - int(g['t'][0]))
    _close(eps, g['eps'], 1e-4)
    if tag == 'tiny':
        den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(100), dev)
        x = den.sample(g['loop_obj_embed'], g['loop_triples'], synth.layout_noise(8, 8, 100, seed=9), n_steps=10)
        _close(x, g['loop_x10'], 2e-4)


@pytest.mark.parametrize('tag,concat', [('crossattn', False), ('concat', True)])
def test_unet1d_model_channels_384_vs_reference_golden(dev, tag, concat):
    """GroupNorm32(32, channels) for channels the rows kernels do not reduce in registers (VERDICT r4 "missing #3"; reference: any
    channels % 32 == 0, ldm_diffusion_util.py:222-239): model_channels = 384 -> groups of 12 / 24 / 36 channels.  Those norms run as
    their own launches over whole matrices (plan.norm_segs -> es_groupnorm_vol, one voxel per row) and the plan carries no K-split
    slab tensors; eps and 3 loop steps against the reference, eager and as a captured graph."""
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    g = load_golden('unet1d_mc384_' + tag)
    kw = dict(escfg.layout_denoiser_kwargs(384, concat=concat))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='unet1d_mc384_%s.' % tag)
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
    eps = den.eps(g['box'], g['obj_embed'], g['triples'], iteration=999 - int(g['t'][0]))
    _close(eps, g['eps'], 1e-4)
    den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(100), dev)
    noise = synth.layout_noise(8, 8, 100, seed=11)
    xs = [den.sample(g['loop_obj_embed'], g['loop_triples'], noise, n_steps=3, use_graph=ug) for ug in (False, True)]
    _close(xs[0], g['loop_x3'], 2e-4)
    assert torch.equal(xs[0], xs[1])


def test_linear_split_k_slab_chain_and_rowsel(dev):
    """Round 3: K split over workgroups.  A product writes S partial-sum slabs (slice 0 carries bias + residual), and whoever
    reads the tensor next sums them in fixed order: (a) as an A segment under a per-segment GroupNorm(+SiLU) prologue next
    to a raw segment, (b) as a residual, (c) under a LayerNorm prologue, (d) with the deferred ReLU (``pre_act``).
    Then the device-step-indexed row select used for the per-schedule time tables."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, View, seg, norm_segs
    rs = np.random.RandomState(5)
    M, K, N = 27, 3 * 512, 512
    X = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32))
    W = torch.from_numpy((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32))
    R = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32))
    ga = torch.from_numpy((1 + 0.1 * rs.standard_normal(N)).astype(np.float32))
    be = torch.from_numpy((0.1 * rs.standard_normal(N)).astype(np.float32))
    W2 = torch.from_numpy((rs.standard_normal((N, 2 * N)) / np.sqrt(2 * N)).astype(np.float32))
    W3 = torch.from_numpy((rs.standard_normal((64, N)) / np.sqrt(N)).astype(np.float32))
    ref1 = F.linear(X, W, bias) + R
    for silu in (False, True):
        b = Builder(dev)
        x = b.dev(X)
        h = b.linear([seg(View(x, col=0, width=512)), seg(View(x, col=512, width=512)), seg(View(x, col=1024, width=512))],
                     PackedLinear(W, bias, dev), M, res=View(b.dev(R)))
        assert h.nslab == 3 and h.slab_stride == M * N, (h.nslab, h.slab_stride)       # round 5: one slice per 512-column segment (at most 4 slabs, 256 workgroups)
        # (K slices never straddle segments since round 5: the same operand as ONE segment cut in two)
        h2 = b.linear([seg(View(x, col=0, width=1536))], PackedLinear(W, bias, dev), M, res=View(b.dev(R)), split=48)
        assert h2.nslab == 2
        # (a) + (b): [GN(+SiLU)(h) | raw x] @ W2 + h
        o = b.linear(norm_segs([h], b.dev(ga), b.dev(be), 1e-5, silu) + [seg(View(x, col=0, width=512))],
                     PackedLinear(W2, None, dev), M, res=h)
        # (c): LayerNorm over the slab tensor, (d): relu(h) as the operand of a plain product
        o_ln = b.linear([seg(h2, pro=hip.PRO_LN, gamma=b.dev(ga), beta=b.dev(be), eps=1e-5, gs=N)], PackedLinear(W3, None, dev), M)
        o_relu = b.linear([seg(h, pre_act=hip.ACT_RELU)], PackedLinear(W3, None, dev), M, act=hip.ACT_RELU)
        b.finish().run()
        torch.cuda.synchronize()
        _close(h.value().cpu(), ref1, 2e-5)
        y = F.group_norm(ref1, 32, ga, be, 1e-5)
        y = F.silu(y) if silu else y
        _close(o.value().cpu(), F.linear(torch.cat([y, X[:, :512]], 1), W2) + ref1, 5e-5)
        _close(o_ln.value().cpu(), F.linear(F.layer_norm(ref1, (N,), ga, be, 1e-5), W3), 5e-5)
        assert o_relu.nslab == 1                                                     # an activation epilogue needs finished sums
        _close(o_relu.value().cpu(), F.relu(F.linear(F.relu(ref1), W3)), 5e-5)
        # the split is a function of (K, N) only: a 300-row batch cuts K in the same places -> identical rows
        b2 = Builder(dev)
        xb = b2.dev(torch.cat([X] * 12)[:300])
        hb = b2.linear([seg(View(xb, col=0, width=512)), seg(View(xb, col=512, width=512)), seg(View(xb, col=1024, width=512))],
                       PackedLinear(W, bias, dev), 300, res=View(b2.dev(torch.cat([R] * 12)[:300])))
        b2.finish().run()
        torch.cuda.synchronize()
        assert hb.nslab == h.nslab and torch.equal(hb.value()[:M].cpu(), h.value().cpu())
    # row select: out[r, :] = table[step, :]
    from echoscene_amd.plan import Builder, View
    b = Builder(dev)
    table = b.dev(torch.from_numpy(rs.standard_normal((7, 300)).astype(np.float32)))
    step = b.buf(1, dtype=torch.int32, zero=True)
    step.fill_(4)
    out = b.buf(5, 320, zero=True)
    b.rowsel(table, step, View(out, col=8, ld=320, width=300), rows=5)
    b.finish().run()
    torch.cuda.synchronize()
    assert torch.equal(out[:, 8:308].cpu(), table[4].cpu().expand(5, 300)) and float(out[:, :8].abs().sum()) == 0.0


@pytest.mark.parametrize('C,M', [(512, 32), (128, 8), (256, 45)])
def test_rows_formed_row_layernorm_prologue(dev, C, M):
    """ES_PRO_LN_ATTN (round 5): the launch forms x = rstd(t0) u + t0 + cav from the producer's [t0 | u] slab tensor (u = W1 P t0
    through folded weights, P = LayerNorm's mean subtraction), publishes it through ``res`` and multiplies LayerNorm(x) by a GEGLU
    projection -- against the unfolded arithmetic x = LN(t0) W1^T + b + t0 + cav in float64 (attention.py:172-219 on one token), incl.
    a row tile with rows past M and rows whose mean is large against their spread."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, View, seg
    import ctypes
    rs = np.random.RandomState(C + M)
    f = lambda *sh: torch.from_numpy(rs.standard_normal(sh).astype(np.float32))
    Xin = f(M, C)
    Wp, bp = f(C, C) / np.sqrt(C), 0.1 * f(C) + 4.0          # rows of t0 with |mean| = 4 x their spread
    W1, b1 = f(C, C) / np.sqrt(C), 0.1 * f(C)
    Wg, bg = f(8 * C, C) / np.sqrt(C), 0.1 * f(8 * C)
    cav = f(M, C)
    W1P = W1.double() - W1.double().mean(dim=1, keepdim=True)
    b = Builder(dev)
    x = b.dev(Xin)
    t0u = b.linear([seg(View(x))], PackedLinear(torch.cat([Wp.double(), W1P @ Wp.double()], 0).float(),
                                                torch.cat([bp.double(), W1P @ bp.double()], 0).float(), dev), M,
                   split=max(8, (C // 16 + 1) // 2))
    assert t0u.nslab <= 2
    t0 = t0u.cols(0, C)
    t2 = View(b.buf(M, C, zero=True))
    cv = b.dev(cav + b1)                                     # (the self-attention's bias rides in the cross-attention vector)
    gl = b.linear([seg(t0, pro=hip.PRO_LN_ATTN, eps=1e-5, gs=C)], PackedLinear(Wg, bg, dev, geglu=True), M, res=t2, res2=View(cv))
    assert hip.lib().es_linear_rows_takes_ln_attn(ctypes.byref(b.ops[-1].u.linear)) == 1
    b.finish().run()
    torch.cuda.synchronize()
    T0 = F.linear(Xin.double(), Wp.double(), bp.double())
    X2 = F.linear(F.layer_norm(T0, (C,), None, None, 1e-5), W1.double(), b1.double()) + T0 + cav.double()
    H = F.linear(F.layer_norm(X2, (C,), None, None, 1e-5), Wg.double(), bg.double())
    ref = H[:, :4 * C] * F.gelu(H[:, 4 * C:])
    _close(t0.value().cpu(), T0.float(), 2e-5)
    _close(t2.value().cpu(), X2.float(), 5e-5)
    _close(gl.value().cpu(), ref.float(), 1e-4)


def test_linear_multi_problem_launch(dev):
    """Independent products marked ``fuse_next`` run as ONE grid (es_linear_rows_multi_f32): different M / K / N / slice counts, a
    gathered operand next to a direct one -- results identical to launching them one by one (ES_ROWS_FUSE semantics)."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, View, seg
    rs = np.random.RandomState(21)
    O, T = 11, 45
    obj = torch.from_numpy(rs.standard_normal((O, 96)).astype(np.float32))
    pred = torch.from_numpy(rs.standard_normal((T, 32)).astype(np.float32))
    si = torch.from_numpy(rs.randint(0, O, T)).long()
    W1 = torch.from_numpy((rs.standard_normal((80, 128)) / 11).astype(np.float32))
    W2 = torch.from_numpy((rs.standard_normal((48, 96)) / 10).astype(np.float32))
    W3 = torch.from_numpy((rs.standard_normal((32, 32)) / 6).astype(np.float32))
    R = torch.from_numpy(rs.standard_normal((T, 32)).astype(np.float32))
    outs = []
    for fuse in (True, False):
        b = Builder(dev)
        o_, p_, s_ = b.dev(obj), b.dev(pred), b.dev(si, torch.int32)
        a1 = b.linear([seg(View(o_), hip.SEG_GATHER, idx=s_, width=96), seg(View(p_))], PackedLinear(W1, None, dev), T, fuse_next=fuse)
        a2 = b.linear([seg(View(o_))], PackedLinear(W2, torch.ones(48), dev), O, fuse_next=fuse)
        a3 = b.linear([seg(View(p_))], PackedLinear(W3, None, dev), T, res=View(b.dev(R)))
        b.finish().run()
        torch.cuda.synchronize()
        outs.append([a1.value().cpu(), a2.value().cpu(), a3.value().cpu()])
    _close(outs[0][0], F.linear(torch.cat([obj[si], pred], 1), W1), 2e-5)
    _close(outs[0][1], F.linear(obj, W2) + 1.0, 2e-5)
    _close(outs[0][2], F.linear(pred, W3) + R, 2e-5)
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)


def test_linear_folded_rider_with_groupnorm_prologue(dev):
    """A node-row product (GroupNorm + SiLU over a SLAB operand, K split) riding on a triple-row launch with more row tiles (8 against
    3): the rider is FOLDED over all rows of the grid (RowsLaunch.fw, its 3 x 6 tiles do not fill the 3 x 8 block: the tail exits) --
    identical to launching the two one after the other, and equal to torch."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, PackedLinear, View, seg, norm_segs
    rs = np.random.RandomState(33)
    T, O, C = 124, 33, 128
    tri = torch.from_numpy(rs.standard_normal((T, 64)).astype(np.float32))
    x0 = torch.from_numpy(rs.standard_normal((O, 256)).astype(np.float32))
    W0 = torch.from_numpy((rs.standard_normal((C, 256)) / 16).astype(np.float32))          # producer of the slab operand (K split)
    Wh = torch.from_numpy((rs.standard_normal((80, 64)) / 8).astype(np.float32))
    Wr = torch.from_numpy((rs.standard_normal((40, C)) / 11).astype(np.float32))
    ga = torch.from_numpy(rs.standard_normal(C).astype(np.float32))
    be = torch.from_numpy(rs.standard_normal(C).astype(np.float32))
    emb = torch.from_numpy(rs.standard_normal((1, 40)).astype(np.float32))
    outs = []
    for fuse in (True, False):
        b = Builder(dev)
        h = b.linear([seg(View(b.dev(x0)))], PackedLinear(W0, None, dev), O, split=8)    # 2 slabs of [O, C]
        assert h.nslab == 2
        a1 = b.linear([seg(View(b.dev(tri)))], PackedLinear(Wh, torch.zeros(80), dev), T, View(b.buf(T, 80)), act=hip.ACT_RELU,
                      fuse_next=fuse)
        a2 = b.linear(norm_segs([h], b.dev(ga), b.dev(be), 1e-5, True, C=C), PackedLinear(Wr, None, dev), O,
                      res=View(b.dev(emb), ld=0, width=40), split=4)
        assert a2.nslab == 2
        b.finish().run()
        torch.cuda.synchronize()
        outs.append([a1.value().cpu(), a2.value().cpu()])
    hh = F.linear(x0, W0)
    _close(outs[0][0], F.relu(F.linear(tri, Wh)), 2e-5)
    _close(outs[0][1], F.linear(F.silu(F.group_norm(hh, 32, ga, be, 1e-5)), Wr) + emb, 1e-4, 1e-4)
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)


def test_unet1d_riding_modes_are_bit_identical(dev):
    """The head of the UNet1D trunk rides on the GCN chain's launches (plan.ROWS_RIDE): eps of one step is BIT-identical with riding
    off (0), on net2's output launches (1) and on net1's / net2's (2) -- only the launch grouping differs."""
    from echoscene_amd import plan, synth, config as escfg
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import LayoutDenoiser
    net = UNet1DModel(**escfg.layout_denoiser_kwargs(128))
    synth.seeded_fill_(net, prefix='ride.')
    O = 12
    _, triples = synth.synthetic_graph(O, seed=5)
    oe = torch.randn(O, 640, generator=torch.Generator().manual_seed(5))
    x = torch.randn(O, 8, generator=torch.Generator().manual_seed(6))
    old, res, nops = plan.ROWS_RIDE, [], []
    try:
        for mode in (0, 1, 2):
            plan.ROWS_RIDE = mode
            den = LayoutDenoiser(net, escfg.layout_diffusion_kwargs(1000), dev)
            res.append(den.eps(x, oe, triples, 3).cpu())
            res.append(den.sample(oe, triples, noise=synth.layout_noise(O, 8, 6), n_steps=6, use_graph=True).cpu())
    finally:
        plan.ROWS_RIDE = old
    assert torch.isfinite(res[0]).all() and float(res[0].abs().max()) > 0
    for k in (2, 4):
        assert torch.equal(res[0], res[k]) and torch.equal(res[1], res[k + 1])


def test_plan_reuse_across_scene_graphs(dev):
    """eval_3dfront.py visits a different scene graph on every call: plans are cached by (node count, triple-row capacity)
    and a new graph of the same size class only rewrites index arrays in place -- same plan object, same captured hipGraph,
    results bit-identical to a denoiser built for that graph (padding rows never reach a result)."""
    den = _layout(dev, 128, 128, 'unet1d_tiny.', 100)
    noise = synth.layout_noise(8, 8, 100, seed=7)
    cases = []
    for seed in (2, 3):
        objs, tri = synth.synthetic_graph(8, seed=seed)
        cases.append((tri, torch.randn(8, 640, generator=torch.Generator().manual_seed(seed))))
    cases.append((cases[1][0][:-3].contiguous(), cases[1][1]))          # fewer triples, same capacity class
    outs, handles = [], []
    for tri, oe in cases:
        outs.append(den.sample(oe, tri, noise, n_steps=20))
        handles.append(den._last['plan'].handle)
    assert len(den._plans) == 1 and len(set(handles)) == 1
    assert torch.equal(den.sample(cases[0][1], cases[0][0], noise, n_steps=20), outs[0])      # and back again
    for (tri, oe), got in zip(cases, outs):
        fresh = _layout(dev, 128, 128, 'unet1d_tiny.', 100)
        assert torch.equal(fresh.sample(oe, tri, noise, n_steps=20), got)
    assert not torch.equal(outs[1], outs[2])
