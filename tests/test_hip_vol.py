"""GPU parity tests of the "volume" path (3-D latent-SDF denoiser), all through the C ABI.

Numerics contract of the path: fp16 MFMA operands, fp32 accumulation, fp32 residual
stream and statistics.  Two references are used:
  * an fp16-operand emulation (PyTorch CPU fp32 ops on operands rounded to fp16, or the oracle with
    ``Numerics(torch.float16)``) -- differs from the kernels only in summation order / online softmax
    -> tight tolerance (1e-3 relative to the tensor scale);
  * the exact fp32 oracle / the reference-generated golden vectors -> the stated precision of the
    fp16 path: 2e-2 relative to the tensor scale (measured values are printed with -s).
"""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, seeded_state_dict
from echoscene_amd import synth, config as escfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda')


def _rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-6)).item()


def _rnd(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32))


def _cl(x):          # NCDHW -> [O*V, C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def _ncdhw(y, O, D, H, W):
    return y.reshape(O, D, H, W, -1).permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize('N,cin,taps,geglu', [(224, 224, 27, False), (672, 1344, 27, False), (3, 224, 27, False), (100, 40, 1, False),
                                              (448, 5, 27, False), (3584, 448, 1, True)])
def test_conv_weight_relayout_on_the_device_equals_the_host_loop(dev, N, cin, taps, geglu):
    """es_pack_conv_f16_dev (what the planner uses) against es_pack_conv_f16: the same tiled, swizzled fp16 image bit for bit
    (padding rows / channels zero, round to nearest even)."""
    from echoscene_amd.plan_vol import PackedConv
    rs = np.random.RandomState(N + cin)
    shape = (N, cin) if taps == 1 else (N, cin, 3, 3, 3)
    W = torch.from_numpy((rs.standard_normal(shape) * rs.choice([1e-6, 1e-3, 1.0, 300.0], shape)).astype(np.float32))   # subnormals .. near overflow
    b = torch.from_numpy(rs.standard_normal(N).astype(np.float32))
    d, h = PackedConv(W, b, dev, geglu=geglu), PackedConv(W, b, 'cpu', geglu=geglu)
    torch.cuda.synchronize()
    assert d.w.is_cuda and d.geglu == h.geglu and torch.equal(d.w.cpu(), h.w) and torch.equal(d.b.cpu(), h.b)


@pytest.mark.parametrize('O,Cin,N,dims,bias', [(3, 224, 3, (16, 16, 16), True), (2, 96, 5, (4, 8, 16), False), (1, 128, 16, (8, 4, 32), True)])
def test_output_conv_narrow_n_kernel(dev, O, Cin, N, dims, bias):
    """The UNet's output conv (out.2: 3x3x3, 224 -> 3, NCDHW fp32; openai_model_3d.py:735-739) on k_conv_n16 (a halo'd LDS image per
    4 x 4 x 16 block + ONE 16-column MFMA fragment per tap) vs F.conv3d on the fp16-rounded operands, and -- same K order, same fp32
    accumulation chain -- BIT-identical to the 224-column tile kernels' result for the same conv (channels-last output route)."""
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    D, H, W = dims
    x = _rnd((O, Cin) + dims, 11).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3), 12) / np.sqrt(Cin * 27)).half().float()
    bs = _rnd((N,), 13) if bias else None
    ref = F.conv3d(x, wt, bs, padding=1)
    b = Builder(dev)
    a16 = b.dev(_cl(x), torch.float16)
    pc = PackedConv(wt, bs, dev)
    out = b.buf(O, N, D, H, W, zero=True)
    b.conv(a16, pc, O, dims, out_f32=out, ncdhw=True)                 # -> k_conv_n16
    Np = (N + 3) // 4 * 4
    wt4 = torch.zeros(Np, Cin, 3, 3, 3)
    wt4[:N] = wt
    bs4 = torch.zeros(Np)
    if bias:
        bs4[:N] = bs
    out_cl = b.buf(O * D * H * W, Np, zero=True)
    b.conv(a16, PackedConv(wt4, bs4, dev), O, dims, out_f32=out_cl, splitk=1)     # channels-last output: the 224-column tile kernels, K not split
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-4
    assert torch.equal(out.cpu(), _ncdhw(out_cl.cpu(), O, D, H, W)[:, :N]), 'k_conv_n16 and the tile kernels must sum in the same order'


@pytest.mark.parametrize('O,dims,Cin,N,geglu', [(4, (16, 8, 8), 448, 448, False), (4, (16, 4, 4), 672, 2016, False), (1, (16, 8, 8), 448, 3584, True),
                                                (3, (4, 4, 4), 160, 250, False)])
def test_small_linear_launches_on_the_deep_ring_kernel(dev, O, dims, Cin, N, geglu):
    """k_linear_deep: small, K-short 1x1 / linear launches (the transformer linears at few objects per GPU) with a 7-slot ring issued at
    kernel entry, no split K, no reduction kernel.  Against torch on the fp16-rounded operands, and BIT-identical to the unsplit launch
    of the ordinary 64-row tile kernel (`splitk=1` keeps the launch off the deep-ring route: same K order, same accumulation chain)."""
    from echoscene_amd import hip
    # (round 6: the few-objects routes take these shapes in the default routing -- k_conv_kw or 64-row producer/consumer tiles,
    #  test_conv_few_objects_kernels_bit_for_bit; the deep-ring kernel is the route of `conv_few = 0`)
    hip.check(hip.lib().es_vol_set_option(b'conv_few', 0), 'es_vol_set_option')
    try:
        _deep_ring_case(dev, O, dims, Cin, N, geglu)
    finally:
        hip.check(hip.lib().es_vol_set_option(b'conv_few', 1), 'es_vol_set_option')


def _deep_ring_case(dev, O, dims, Cin, N, geglu):
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    D, H, W = dims
    M = O * D * H * W
    x = _rnd((M, Cin), 1).half().float()
    wt = (_rnd((N, Cin), 2) / np.sqrt(Cin)).half().float()
    bias = _rnd((N,), 3)
    b = Builder(dev)
    a16 = b.dev(x, torch.float16)
    if geglu:
        pc = PackedConv(wt, bias, dev, geglu=True)
        assert pc.geglu
        h = x @ wt.t() + bias
        ref = h[:, :N // 2] * F.gelu(h[:, N // 2:])
        outs = [b.buf(M, N // 2, dtype=torch.float16, zero=True) for _ in range(2)]
        b.conv(a16, pc, O, dims, out_f16=outs[0], epilogue=hip.EPI_GEGLU, out_ld=N // 2)
        i1 = b.conv(a16, pc, O, dims, out_f16=outs[1], epilogue=hip.EPI_GEGLU, out_ld=N // 2)
        b.ops[i1].u.conv.splitk = 1                  # (es_conv_args.splitk = 1: "no split, ordinary route")
        b.finish().run()
        torch.cuda.synchronize()
        assert _rel(outs[0], ref) < 2e-3
        assert torch.equal(outs[0], outs[1])
        return
    pc = PackedConv(wt, bias, dev)
    rowv, res = _rnd((O, N), 4), _rnd((M, N), 5)
    ref = x @ wt.t() + bias + rowv.repeat_interleave(D * H * W, 0) + res
    o32 = [b.buf(M, N, zero=True) for _ in range(2)]
    o16 = [b.buf(M, N, dtype=torch.float16, zero=True) for _ in range(2)]
    rv, rs = View(b.dev(rowv)), b.dev(res)
    b.conv(a16, pc, O, dims, rowvec=rv, res=rs, out_f32=o32[0], out_f16=o16[0])
    b.conv(a16, pc, O, dims, rowvec=rv, res=rs, out_f32=o32[1], out_f16=o16[1], splitk=1)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(o32[0], ref) < 1e-4
    assert torch.equal(o32[0], o32[1]) and torch.equal(o16[0], o16[1])


@pytest.mark.parametrize('mode,N,Cin,dims', [('same', 40, 32, (4, 8, 8)), ('same', 224, 64, (4, 8, 8)),
                                              ('same', 250, 96, (2, 4, 4)), ('down', 48, 64, (4, 4, 4)),
                                              ('up', 48, 32, (4, 8, 8)), ('lin', 300, 64, (4, 4, 4))])
def test_conv_mfma(dev, mode, N, Cin, dims):
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    O = 3
    D, H, W = dims
    taps = 1 if mode == 'lin' else 27
    idims = dict(same=dims, lin=dims, down=(D, 2 * H, 2 * W), up=(D, H // 2, W // 2))[mode]
    x = _rnd((O, Cin) + idims, 1).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3) if taps == 27 else (N, Cin), 2) / np.sqrt(Cin * taps)).half().float()
    bias = _rnd((N,), 3)
    rowv = _rnd((O, N), 4)
    res = _rnd((O * D * H * W, N), 5)
    if mode == 'same':
        ref = F.conv3d(x, wt, bias, padding=1)
    elif mode == 'down':
        ref = F.conv3d(x, wt, bias, stride=(1, 2, 2), padding=1)
    elif mode == 'up':
        ref = F.conv3d(F.interpolate(x, (D, H, W), mode='nearest'), wt, bias, padding=1)
    else:
        ref = F.conv3d(x, wt[:, :, None, None, None], bias)
    ref = _cl(ref) + rowv.repeat_interleave(D * H * W, 0) + res
    b = Builder(dev)
    pc = PackedConv(wt, bias, dev)
    a16 = b.dev(_cl(x), torch.float16)
    out = b.buf(O * D * H * W, N, zero=True)
    out16 = b.buf(O * D * H * W, N, dtype=torch.float16, zero=True)
    b.conv(a16, pc, O, dims, mode=dict(same=0, lin=0, down=1, up=2)[mode], rowvec=View(b.dev(rowv)),
           res=b.dev(res), out_f32=out, out_f16=out16)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-4
    assert _rel(out16, ref) < 2e-3


@pytest.mark.parametrize('taps,Cin,Cs,N,dims,O', [(27, 96, 0, 224, (4, 8, 8), 3), (27, 64, 96, 250, (4, 4, 8), 3), (1, 448, 0, 448, (4, 4, 8), 5),
                                                  (1, 160, 64, 72, (2, 4, 8), 3)])
def test_conv_few_objects_kernels_bit_for_bit(dev, taps, Cin, Cs, N, dims, O):
    """Round 6, the few-objects kernels.  (a) producer/consumer tiles of 64 / 128 rows: the unsplit launch has the K order and the
    MFMA chain of every other conv kernel -> the same bits as the dispatcher's own unsplit route.  (b) k_conv_kw, K split INSIDE the
    workgroup over KS streams: stream s multiplies the s-th range of split_range(KS) and the streams are added in stream order -> the
    same bits as a split of S = KS over workgroups followed by k_conv_splitk_reduce.  Ragged rows, fused 1x1 skip phase, per-object
    vector, residual, f16 copy; vs the fp32 reference as well."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    D, H, W = dims
    M = O * D * H * W
    x = _rnd((O, Cin) + dims, 1).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3) if taps == 27 else (N, Cin), 2) / np.sqrt(Cin * taps)).half().float()
    bias, rowv, res = _rnd((N,), 3), _rnd((O, N), 4), _rnd((M, N), 5)
    ref = F.conv3d(x, wt if taps == 27 else wt[:, :, None, None, None], bias, padding=1 if taps == 27 else 0)
    skip = None
    if Cs:
        xs = _rnd((O, Cs) + dims, 6).half().float()
        ws = (_rnd((N, Cs), 7) / np.sqrt(Cs)).half().float()
        ref = ref + F.conv3d(xs, ws[:, :, None, None, None])
    ref = _cl(ref) + rowv.repeat_interleave(D * H * W, 0) + res
    lib = hip.lib()

    def run(opts, splitk):
        for k, v in opts.items():
            hip.check(lib.es_vol_set_option(k.encode(), v), 'es_vol_set_option')
        try:
            b = Builder(dev)
            o32, o16 = b.buf(M, N, zero=True), b.buf(M, N, dtype=torch.float16, zero=True)
            sk = (b.dev(_cl(xs), torch.float16), PackedConv(ws, None, dev)) if Cs else None
            b.conv(b.dev(_cl(x), torch.float16), PackedConv(wt, bias, dev), O, dims, rowvec=View(b.dev(rowv)), res=b.dev(res),
                   out_f32=o32, out_f16=o16, skip=sk, splitk=splitk)
            b.finish().run()
            torch.cuda.synchronize()
            return o32.clone(), o16.clone()
        finally:
            for k in opts:
                hip.check(lib.es_vol_set_option(k.encode(), 4 if k == 'conv_st_np' else 0), 'es_vol_set_option')

    plain = {S: run({}, S) for S in (1, 2, 3, 4, 8)}
    assert _rel(plain[1][0], ref) < 1e-4
    # the dispatcher's own choice for this small problem (few-objects routes: 128-row tiles -- with the shared A tile of k_conv_ws3 for
    # 3x3x3 SAME launches --, K streams, 64-row tiles) is SOME plain split, bit for bit
    auto = run({}, None)
    assert any(torch.equal(auto[0], plain[S][0]) and torch.equal(auto[1], plain[S][1]) for S in plain), 'auto route matches no plain split'
    for opts in ({'conv_st_bm': 64}, {'conv_st_bm': 128}, {'conv_st_bm': 128, 'conv_st_np': 8}):
        for S in (1, 2):
            got = run(opts, S)
            assert torch.equal(got[0], plain[S][0]) and torch.equal(got[1], plain[S][1]), (opts, S)
    for ks in (4, 2):
        got = run({'conv_kw_ks': ks}, 1)
        assert _rel(got[0], ref) < 1e-4
        assert torch.equal(got[0], plain[ks][0]) and torch.equal(got[1], plain[ks][1]), ks
        got2 = run({'conv_kw_ks': ks}, 2)             # a split over workgroups on top: partial slabs of pre-summed streams
        assert _rel(got2[0], ref) < 1e-4


def test_conv_fused_skip_and_ncdhw(dev):
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    O, dims, Cin, Cs, N = 2, (4, 4, 4), 64, 96, 72
    D, H, W = dims
    x = _rnd((O, Cin) + dims, 1).half().float()
    xs = _rnd((O, Cs) + dims, 2).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3), 3) / np.sqrt(Cin * 27)).half().float()
    ws = (_rnd((N, Cs), 4) / np.sqrt(Cs)).half().float()
    bias = _rnd((N,), 5)
    ref = _cl(F.conv3d(x, wt, bias, padding=1) + F.conv3d(xs, ws[:, :, None, None, None]))
    b = Builder(dev)
    out = b.buf(O * D * H * W, N, zero=True)
    b.conv(b.dev(_cl(x), torch.float16), PackedConv(wt, bias, dev), O, dims,
           skip=(b.dev(_cl(xs), torch.float16), PackedConv(ws, None, dev)), out_f32=out)
    # final-conv form: 3 output channels written as NCDHW
    w3 = (_rnd((3, Cin, 3, 3, 3), 6) / np.sqrt(Cin * 27)).half().float()
    out3 = b.buf(O, 3, D, H, W, zero=True)
    b.conv(b.dev(_cl(x), torch.float16), PackedConv(w3, None, dev), O, dims, out_f32=out3, ncdhw=True)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-4
    assert _rel(out3, F.conv3d(x, w3, None, padding=1)) < 1e-4


def test_groupnorm_layernorm_geglu_tocl(dev):
    from echoscene_amd.plan import Builder
    O, C1, C2, dims = 3, 64, 32, (4, 4, 8)
    V = dims[0] * dims[1] * dims[2]
    x1, x2 = _rnd((O, C1) + dims, 1) * 2 + 0.5, _rnd((O, C2) + dims, 2)
    ga, be = 1 + 0.1 * _rnd((C1 + C2,), 3), 0.1 * _rnd((C1 + C2,), 4)
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 1), 32, ga, be, 1e-5))
    b = Builder(dev)
    y = b.buf(O * V, C1 + C2, dtype=torch.float16, zero=True)
    raw = b.buf(O * V, C1 + C2, dtype=torch.float16, zero=True)
    b.groupnorm(b.dev(_cl(x1)), C1, b.dev(_cl(x2)), C2, O, V, b.dev(ga), b.dev(be), 1e-5, True, y, raw)
    y1 = b.buf(O * V, C1, dtype=torch.float16, zero=True)
    b.groupnorm(b.dev(_cl(x1)), C1, None, 0, O, V, b.dev(ga[:C1]), b.dev(be[:C1]), 1e-6, False, y1)
    M, Cc = 50, 96
    t = _rnd((M, Cc), 5) * 3 - 1
    yl = b.buf(M, Cc, dtype=torch.float16, zero=True)
    b.layernorm(b.dev(t), M, Cc, b.dev(ga), b.dev(be), yl)
    hg = _rnd((M, 2 * 64), 6)
    yg = b.buf(M, 64, dtype=torch.float16, zero=True)
    b.geglu(b.dev(hg), M, 64, yg)
    xin = _rnd((O, 3) + dims, 7)
    xcl = b.buf(O * V, 32, dtype=torch.float16)
    b.to_cl(b.dev(xin), O, 3, V, 32, xcl)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(y, _cl(ref)) < 2e-3
    assert _rel(raw, _cl(torch.cat([x1, x2], 1))) < 1e-3
    assert _rel(y1, _cl(F.group_norm(x1, 32, ga[:C1], be[:C1], 1e-6))) < 2e-3
    assert _rel(yl, F.layer_norm(t, (Cc,), ga, be, 1e-5)) < 2e-3
    a_, g_ = hg.chunk(2, -1)
    assert _rel(yg, a_ * F.gelu(g_)) < 2e-3
    assert _rel(xcl[:, :3], _cl(xin)) < 1e-3 and xcl[:, 3:].abs().max() == 0


@pytest.mark.parametrize('M,Cc', [(9001, 448), (8192, 672), (37, 1024), (3, 8), (20000, 100)])
def test_layernorm_tokens_row_loop(dev, M, Cc):
    """k_layernorm (round 5: resident waves loop over the rows, the next row's loads in flight): row counts above / below the
    launch's wave count and not a multiple of it, widths with 1-4 chunks per lane and partial last chunks; f16 and fp32 outputs."""
    from echoscene_amd.plan import Builder
    t = _rnd((M, Cc), 11) * 3 - 1
    ga, be = 1 + 0.1 * _rnd((Cc,), 12), 0.1 * _rnd((Cc,), 13)
    ref = F.layer_norm(t, (Cc,), ga, be, 1e-5)
    b = Builder(dev)
    y16 = b.buf(M + 1, Cc, dtype=torch.float16, zero=True)
    y32 = b.buf(M + 1, Cc, zero=True)
    x = b.dev(t)
    b.layernorm(x, M, Cc, b.dev(ga), b.dev(be), y16)
    b.layernorm(x, M, Cc, b.dev(ga), b.dev(be), y32)
    b.finish().run()
    torch.cuda.synchronize()
    assert (y32[:M].cpu() - ref).abs().max() < 2e-5
    assert (y16[:M].float().cpu() - ref).abs().max() < 4e-3
    assert float(y16[M].abs().max()) == 0.0 and float(y32[M].abs().max()) == 0.0        # nothing past the last row


def test_gelu_of_the_volume_path_over_its_whole_range(dev):
    """es_gelu_fast (erfc by Abramowitz-Stegun 7.1.26 on the non-cancelling side; FeedForward GEGLU, reference attention.py:39-46
    uses F.gelu = exact erf): gate values on a grid over [-12, 12] with value 1 -> the fp16 output IS gelu(gate).  Bound: one fp16
    rounding of the exact value plus 1e-6 absolute (the approximation's own error is 2.2e-7)."""
    from echoscene_amd.plan import Builder
    M, C = 64, 256
    gate = torch.linspace(-12, 12, M * C).reshape(M, C)
    hg = torch.cat([torch.ones(M, C), gate], 1).contiguous()
    b = Builder(dev)
    y = b.buf(M, C, dtype=torch.float16, zero=True)
    b.geglu(b.dev(hg), M, C, y)
    b.finish().run()
    torch.cuda.synchronize()
    ref = F.gelu(gate.double())
    err = (y.double().cpu() - ref).abs()
    bound = ref.abs() * 2.0 ** -10 + 1e-6 + 6e-8
    assert bool((err <= bound).all()), 'max excess %.3e' % (err - bound).max().item()


@pytest.mark.parametrize('Ntok,heads,dh', [(256, 8, 12), (1024, 2, 56), (256, 2, 84), (100, 3, 8)])
def test_attention(dev, Ntok, heads, dh):
    from echoscene_amd.plan import Builder
    import echoscene_amd.plan_vol  # noqa: F401  (attaches the volume ops to Builder)
    B, Cc = 2, heads * dh
    qkv = (_rnd((B, Ntok, 3 * Cc), 1) * 1.5).half()
    q, k, v = qkv.float().chunk(3, -1)
    sp = lambda t: t.reshape(B, Ntok, heads, dh).permute(0, 2, 1, 3)
    att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5, -1) @ sp(v)
    ref = att.permute(0, 2, 1, 3).reshape(B * Ntok, Cc)
    b = Builder(dev)
    out = b.buf(B * Ntok, Cc, dtype=torch.float16, zero=True)
    b.attention(b.dev(qkv.reshape(B * Ntok, 3 * Cc), torch.float16), B, Ntok, heads, dh, out)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 3e-3


def test_shape_stem(dev):
    from echoscene_amd.plan import Builder
    from oracle import echoscene_oracle as orc
    O = 3
    x = _rnd((O, 3, 16, 16, 16), 1)
    sd = {'shape_embeddings.0.weight': _rnd((32, 3, 3, 3, 3), 2) / 9, 'shape_embeddings.0.bias': _rnd((32,), 3),
          'shape_embeddings.2.weight': _rnd((64, 32, 3, 3, 3), 4) / 29, 'shape_embeddings.2.bias': _rnd((64,), 5),
          'shape_embeddings.5.weight': torch.eye(512), 'shape_embeddings.5.bias': torch.zeros(512)}
    ref = orc.shape_stem(sd, x)
    b = Builder(dev)
    out = b.buf(O, 512, zero=True)
    b.stem(b.dev(x), [b.dev(sd['shape_embeddings.%d.%s' % (i, n)]) for i in (0, 2) for n in ('weight', 'bias')],
           b.buf(O, 32 * 512), out, O)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-5


def _shape(dev, mc, ctx, prefix, S, precision='fp16', on_gpu=False):
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    p = escfg.shape_unet_params(mc)
    p['context_dim'] = ctx
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix=prefix)
    if on_gpu:
        df.to(dev)
    return ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=S, device=dev, precision=precision)


def _unet3d_sd(den):
    return {k[len('diffusion_net.'):]: v.detach().cpu() for k, v in den.df.state_dict().items()}


def test_unet3d_tiny_blockwise_vs_oracle(dev):
    """Every block output of the tiny 3-D denoiser against the fp16-operand oracle."""
    from oracle import echoscene_oracle as orc
    g = load_golden('unet3d_tiny')
    den = _shape(dev, 32, 64, 'unet3d_tiny.', 100)
    t = int(g['t'][0])
    it = int(np.nonzero(den.sched.timesteps == t)[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it)
    st = next(iter(den._plans.values()))
    trace = {}
    ref = orc.unet3d_forward(_unet3d_sd(den), g['x'], g['uc_s'], g['triples'], g['t'],
                             nm=orc.Numerics(torch.float16), trace=trace)
    bad = []
    for name, v in st['eps_plan'].tags.items():
        if name not in trace:
            continue
        r = trace[name]
        if r.dim() == 5:
            r = r.permute(0, 2, 3, 4, 1).reshape(-1, r.shape[1])
        elif r.dim() == 3:
            r = r.reshape(-1, r.shape[-1])
        got = v.t.reshape(-1, v.t.shape[-1])[:, v.col:v.col + v.width]
        e = _rel(got, r)
        if not e < 2e-3:
            bad.append((name, '%.2e' % e))
    assert not bad, bad[:8]
    assert _rel(eps, ref) < 2e-3


def test_unet3d_tiny_eps_vs_reference_golden(dev):
    g = load_golden('unet3d_tiny')
    den = _shape(dev, 32, 64, 'unet3d_tiny.', 100)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it)
    e = _rel(eps, g['eps'])
    print('unet3d tiny: fp16-MFMA eps vs fp32 reference golden: rel err %.3e' % e)
    assert e < 2e-2


def test_shape_denoiser_from_a_model_on_the_gpu(dev):
    """samplers.state_dict_for: parameters that already live on the GPU are folded (fp64, es_matmul_f64) and re-laid out there; same
    golden as the host route, and the plan keeps no alias of a parameter."""
    g = load_golden('unet3d_tiny')
    den = _shape(dev, 32, 64, 'unet3d_tiny.', 100, on_gpu=True)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it).clone()
    assert _rel(eps, g['eps']) < 2e-2
    host = _shape(dev, 32, 64, 'unet3d_tiny.', 100).eps(g['x'], g['uc_s'], g['triples'], iteration=it)
    assert _rel(eps, host) < 1e-4                      # (fp64 folds in another summation order: a weight may round the other way)
    with torch.no_grad():
        for p_ in den.df.parameters():
            p_.zero_()
    torch.cuda.synchronize()
    assert torch.equal(den.eps(g['x'], g['uc_s'], g['triples'], iteration=it), eps)


@pytest.mark.parametrize('use_graph', [False, True])
def test_ddim_tiny_loop_vs_reference_golden(dev, use_graph):
    g = load_golden('ddim_tiny')
    den = _shape(dev, 32, 64, 'unet3d_tiny.', 4)
    z = den.sample(g['uc_s'], g['triples'], synth.shape_noise(seed=7), use_graph=use_graph)
    e = _rel(z, g['z_final'])
    print('ddim tiny (4 steps): latent vs fp32 reference golden: rel err %.3e' % e)
    assert e < 2e-2
    z2 = den.sample(g['uc_s'], g['triples'], synth.shape_noise(seed=7), use_graph=use_graph)
    assert torch.equal(z, z2)
    # scratch claim of the plan (activations, split-K slabs, statistics are stored empty in model files): NaN-poisoned, same bits
    assert den._plan_for(g['uc_s'], g['triples'], None)['plan'].poison_scratch() > 0
    z3 = den.sample(g['uc_s'], g['triples'], synth.shape_noise(seed=7), use_graph=use_graph)
    assert torch.equal(z, z3), 'an op reads scratch bytes that no op of the plan wrote'


def test_ddim_loop_with_eta_vs_reference_golden(dev):
    """Stochastic DDIM (eta = 0.7: + sigma_t * randn per step and object, samplers/ddim.py:256-260) against the reference's own
    DDIMSampler with the per-step draws injected (make_golden.py case_sampler_options); eta = 0 afterwards must still reproduce the
    deterministic golden (the schedule with four coefficients and no noise table)."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    g = load_golden('sampler_options_tiny')
    p = escfg.shape_unet_params(32)
    p['context_dim'] = 64
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix='unet3d_tiny.')
    den = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev, ddim_eta=0.7)
    assert tuple(den.coef.shape) == (4, 5)
    step_noise = torch.stack([_rnd((4, 3, 16, 16, 16), 900 + k) for k in range(4)])
    z = den.sample(g['ddim_uc_s'], g['ddim_triples'], synth.shape_noise(seed=7), step_noise=step_noise)
    e = _rel(z, g['ddim_z_final_eta07'])
    print('ddim tiny, eta 0.7 (4 steps): latent vs fp32 reference golden: rel err %.3e' % e)
    assert e < 2e-2
    z2 = den.sample(g['ddim_uc_s'], g['ddim_triples'], synth.shape_noise(seed=7), step_noise=step_noise)
    assert torch.equal(z, z2)
    # the model file of a stochastic loop names its draws: region "step_noise" [S, objects x latent] is an input the host fills
    import ctypes as C, tempfile
    from echoscene_amd import hip
    L = hip.lib()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'eta.esm')
        den.save_model(path, g['ddim_uc_s'], g['ddim_triples'])
        m = L.es_model_load(path.encode())
        assert m, L.es_last_error()
        try:
            reg = {}
            for name in (b'x', b'step_noise'):
                ptr, nb = C.c_void_p(), C.c_size_t()
                hip.check(L.es_model_region(C.c_void_p(m), name, C.byref(ptr), C.byref(nb)), 'es_model_region')
                reg[name] = (ptr.value, nb.value)
            O_ = step_noise.shape[1]
            assert reg[b'step_noise'][1] == step_noise.numel() * 4 and reg[b'x'][1] == O_ * 3 * 16 ** 3 * 4
            sn = step_noise.to(dev).float().reshape(4, -1).contiguous()
            x0 = synth.shape_noise(seed=7).to(dev).float().expand(O_, 3, 16, 16, 16).contiguous()
            from echoscene_amd.plan import Builder

            def dcopy(dst, src, nbytes):               # device-to-device copy through the library (raw pointers: ES_OP_COPY)
                bb = Builder(dev)
                bb.copy(dst, src, nbytes)
                bb.finish().run()
                torch.cuda.synchronize()
            dcopy(reg[b'step_noise'][0], sn.data_ptr(), reg[b'step_noise'][1])
            dcopy(reg[b'x'][0], x0.data_ptr(), reg[b'x'][1])
            hip.check(L.es_model_run(C.c_void_p(m), 0, 4, hip.current_stream()), 'es_model_run')
            torch.cuda.synchronize()
            zf = torch.empty_like(x0)
            dcopy(zf.data_ptr(), reg[b'x'][0], reg[b'x'][1])
            assert torch.equal(zf.cpu(), z.cpu()), 'the loaded model replays the stochastic loop bit for bit'
        finally:
            L.es_model_free(C.c_void_p(m))
    z0 = _shape(dev, 32, 64, 'unet3d_tiny.', 4).sample(g['ddim_uc_s'], g['ddim_triples'], synth.shape_noise(seed=7))
    assert _rel(z0, load_golden('ddim_tiny')['z_final']) < 2e-2


def test_unet3d_full_eps_vs_reference_golden(dev):
    """Full-width shape denoiser (config/sdfusion-txt2shape_mp.yaml), O=2."""
    g = load_golden('unet3d_full')
    den = _shape(dev, 224, 1280, 'unet3d_full.', 100)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it)
    e = _rel(eps, g['eps'])
    print('unet3d full: fp16-MFMA eps vs fp32 reference golden: rel err %.3e' % e)
    assert e < 2e-2


@pytest.mark.parametrize('mode,N,Cin,dims,skipC', [('same', 40, 48, (4, 8, 8), 0), ('same', 224, 64, (4, 8, 8), 32), ('down', 48, 64, (4, 4, 4), 0),
                                                    ('up', 72, 32, (4, 8, 8), 0), ('lin', 300, 80, (4, 4, 4), 0), ('same', 3, 64, (4, 4, 4), 0)])
def test_conv_fp32_operand_route(dev, mode, N, Cin, dims, skipC):
    """es_conv_f32 (fp32 activations / weights on v_mfma_f32_16x16x4_f32, csrc/es_vol32.hip): every conv mode of the volume path, the
    fused 1x1 skip, bias + per-object vector + residual, NCDHW output -- against F.conv3d in fp32 on UNROUNDED operands, 1e-5."""
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv32
    O = 3
    D, H, W = dims
    taps = 1 if mode == 'lin' else 27
    idims = dict(same=dims, lin=dims, down=(D, 2 * H, 2 * W), up=(D, H // 2, W // 2))[mode]
    x = _rnd((O, Cin) + idims, 1)
    wt = _rnd((N, Cin, 3, 3, 3) if taps == 27 else (N, Cin), 2) / np.sqrt(Cin * taps)
    bias, rowv, res = _rnd((N,), 3), _rnd((O, N), 4), _rnd((O * D * H * W, N), 5)
    if mode == 'same':
        ref = F.conv3d(x, wt, bias, padding=1)
    elif mode == 'down':
        ref = F.conv3d(x, wt, bias, stride=(1, 2, 2), padding=1)
    elif mode == 'up':
        ref = F.conv3d(F.interpolate(x, (D, H, W), mode='nearest'), wt, bias, padding=1)
    else:
        ref = F.conv3d(x, wt[:, :, None, None, None], bias)
    b = Builder(dev)
    b.fp32 = True
    a32 = b.dev(_cl(x))
    skip = None
    if skipC:
        xs, ws = _rnd((O, skipC) + dims, 6), _rnd((N, skipC), 7) / np.sqrt(skipC)
        ref = ref + F.conv3d(xs, ws[:, :, None, None, None])
        skip = (b.dev(_cl(xs)), PackedConv32(ws, None, dev))
    if N == 3:                                   # the output conv's form: NCDHW, no fusions
        out = b.buf(O, N, D, H, W, zero=True)
        b.conv(a32, PackedConv32(wt, bias, dev), O, dims, out_f32=out, ncdhw=True)
        b.finish().run()
        torch.cuda.synchronize()
        assert _rel(out, ref) < 1e-5
        return
    ref = _cl(ref) + rowv.repeat_interleave(D * H * W, 0) + res
    out = b.buf(O * D * H * W, N, zero=True)
    b.conv(a32, PackedConv32(wt, bias, dev), O, dims, mode=dict(same=0, lin=0, down=1, up=2)[mode], rowvec=View(b.dev(rowv)),
           res=b.dev(res), out_f32=out, skip=skip)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize('precision', ['fp32', 'fp32x'])
@pytest.mark.parametrize('tag,mc,ctx', [('tiny', 32, 64), ('full', 224, 1280)])
def test_unet3d_eps_fp32_operand_route_vs_reference_golden(dev, tag, mc, ctx, precision):
    """the reference is fp32 everywhere (openai_model_3d.py:816-863); ShapeDenoiser(precision='fp32') runs the SAME plan
    with fp32 operands on the exact-fp32 matrix instruction.  One UNet3D + echo-GCN evaluation against the reference golden at the
    fp32 bar: <= 1e-4 relative (the fp16-operand product route measures 8e-4 ... 1.4e-3 on the same goldens).
    Round 6, 'fp32x': the same bar for the SPLIT-OPERAND route -- fp32 activations, every contraction as three f16 partial products
    (hi x hi + lo x hi + hi x lo, fp32 accumulate) on the product kernels."""
    g = load_golden('unet3d_' + tag)
    den = _shape(dev, mc, ctx, 'unet3d_%s.' % tag, 100, precision=precision)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it)
    e = _rel(eps, g['eps'])
    den16 = _shape(dev, mc, ctx, 'unet3d_%s.' % tag, 100)
    e16 = _rel(den16.eps(g['x'], g['uc_s'], g['triples'], iteration=it), g['eps'])
    print('unet3d %s eps vs fp32 reference golden: %s route rel err %.3e, fp16-operand route %.3e' % (tag, precision, e, e16))
    assert e < 1e-4


def test_unet3d_full_eps_O32_vs_reference_golden(dev):
    """The BENCHMARKED routes against the reference: model_channels 224, O = 32 -- every 3x3x3 / 1x1 launch has
    >= 256 row tiles, i.e. the plain k_conv_ws route, the column-panel order and k_linear_ws at 16-24 column tiles that
    bench.py times.  Golden = ONE reference UNet3D + echo-GCN evaluation on CPU (make_golden.py: unet3d_full_O32)."""
    g = load_golden('unet3d_full_O32')
    O = 32
    x = _rnd((O, 3, 16, 16, 16), int(g['x_seed']))
    den = _shape(dev, 224, 1280, 'unet3d_full.', 100)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(x, g['uc_s'], g['triples'], iteration=it)
    e = _rel(eps, g['eps'])
    rms = ((eps.cpu() - g['eps']).pow(2).mean().sqrt() / g['eps'].pow(2).mean().sqrt()).item()
    print('unet3d full O=32: fp16-MFMA eps vs fp32 reference golden: max rel err %.3e, rel rms %.3e' % (e, rms))
    assert e < 2e-2


@pytest.mark.parametrize('mc,ctx,prefix,world,O', [(32, 64, 'unet3d_tiny.', 2, 4), (224, 1280, 'unet3d_full.', 4, 4),
                                                    (224, 1280, 'unet3d_full.', 8, 32), (224, 1280, 'unet3d_full.', 2, 32),
                                                    (224, 1280, 'unet3d_full.', 3, 10)])
def test_object_shards_equal_unsharded_bitwise(dev, mc, ctx, prefix, world, O):
    """the sharded result must equal the single-GPU result BIT FOR BIT.  The multi-GPU decomposition
    on one GPU: ``world`` shards stepped with a simulated all-gather.  Every kernel treats objects independently; what can
    differ is the fp32 summation order, when split-K factors and GroupNorm partial-sum tiles are picked from the LOCAL object
    count.  ``deterministic=True`` (round 6): every rank of every world size -- 1 included -- takes the cuts of a 4-object
    reference shard (es_conv_args.O_hint / es_gn_args.O_hint < 0), all conv kernels cut split-K ranges in the same places and K
    streams inside a workgroup are added in the order of the reduction kernel, so neither the tile nor the kernel chosen per
    launch matters: 32 objects over 8 ranks (4 each: the reference itself), over 2 ranks (16 each: the 256-row tiles take the
    reference's splits), 10 objects over 3 ranks (4 + 4 + 2)."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    objs, triples = synth.synthetic_graph(O, seed=6)
    uc = _rnd((O, 1, ctx), 52)
    noise1 = synth.shape_noise(seed=7)
    p = escfg.shape_unet_params(mc)
    p['context_dim'] = ctx
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix=prefix)
    mpar = escfg.shape_df_conf().model.params
    nst = 4 if O <= 4 else 2                 # (the benchmarked decomposition: 32 objects over 8 ranks, full width)
    z_ref = ShapeDenoiser(df, mpar, ddim_steps=4, device=dev, deterministic=True).sample(uc, triples, noise1, n_steps=nst)
    shards = [ShapeDenoiser(df, mpar, ddim_steps=4, device=dev, rank=r, world=world, deterministic=True) for r in range(world)]
    for sh in shards:
        st = sh._plan_for(uc, triples)
        st['x'].copy_(noise1.to(dev).expand(st['hi'] - st['lo'], 3, 16, 16, 16))
        sh._cur, sh._use_graph = st, True
    for i in range(nst):
        codes = torch.cat([sh.codes_local(i)[:sh._cur['hi'] - sh._cur['lo']].clone() for sh in shards], 0)
        for sh in shards:
            sh.step(i, codes)
    z = torch.cat([sh.latents_local() for sh in shards], 0)
    assert torch.equal(z, z_ref), 'max abs diff %.3e' % (z - z_ref).abs().max().item()


def _run_shards(shards, uc, triples, noise1, nst, dev):
    for sh in shards:
        st = sh._plan_for(uc, triples)
        st['x'].copy_(noise1.to(dev).expand(st['hi'] - st['lo'], 3, 16, 16, 16))
        sh._cur, sh._use_graph = st, True
    for i in range(nst):
        codes = torch.cat([sh.codes_local(i)[:sh._cur['hi'] - sh._cur['lo']].clone() for sh in shards], 0)
        for sh in shards:
            sh.step(i, codes)
    return torch.cat([sh.latents_local() for sh in shards], 0)


@pytest.mark.parametrize('world', [2, 8])
def test_tuned_object_shards_vs_unsharded_run(dev, world):
    """``ShapeDenoiser(deterministic=False)`` -- the mode the strong-scaling figures quote: every rank picks its K
    splits / tiles from its LOCAL object count -- had no numeric test.  O = 32 at the shipped widths over 2 and 8 shards on one
    GPU, 2 DDIM steps (the decomposition of ``test_object_shards_equal_unsharded_bitwise``): equal to the unsharded run up to the
    fp32 summation order of fp16-operand products -- stated bar allclose(2e-2, 2e-2), the measured maximum is printed."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    O, mc, ctx = 32, 224, 1280
    objs, triples = synth.synthetic_graph(O, seed=6)
    uc = _rnd((O, 1, ctx), 52)
    noise1 = synth.shape_noise(seed=7)
    p = escfg.shape_unet_params(mc)
    p['context_dim'] = ctx
    df = DiffusionUNet(p)
    synth.seeded_fill_(df, prefix='unet3d_full.')
    mpar = escfg.shape_df_conf().model.params
    z_ref = ShapeDenoiser(df, mpar, ddim_steps=4, device=dev).sample(uc, triples, noise1, n_steps=2)
    shards = [ShapeDenoiser(df, mpar, ddim_steps=4, device=dev, rank=r, world=world, deterministic=False) for r in range(world)]
    z = _run_shards(shards, uc, triples, noise1, 2, dev)
    d = (z - z_ref).abs().max().item()
    rms = ((z - z_ref).pow(2).mean().sqrt() / z_ref.pow(2).mean().sqrt()).item()
    print('tuned shards, world %d, O = 32, 2 DDIM steps vs the unsharded run: max abs diff %.3e (|z| max %.2f), rel rms %.3e'
          % (world, d, z_ref.abs().max().item(), rms))
    assert torch.allclose(z, z_ref, atol=2e-2, rtol=2e-2), d


def test_tuned_object_shards_vs_reference_trajectory_O16(dev):
    """The tuned shard mode against the REFERENCE: ``shape_traj_full_O16`` (two steps of the reference's own DDIMSampler at
    model_channels 224, 16 objects) stepped as 8 tuned shards of 2 objects -- the kernel routes of a few-objects-per-GPU rank
    (64-row tiles, K-short linear launches, local split-K factors).  Bar: the product path's fp16-operand tolerance."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    g = load_golden('shape_traj_full_O16')
    df = DiffusionUNet(escfg.shape_unet_params(224))
    synth.seeded_fill_(df, prefix='unet3d_full.')
    mpar = escfg.shape_df_conf().model.params
    noise1 = synth.shape_noise(seed=7)
    world = 8
    for k in (1, 2):
        shards = [ShapeDenoiser(df, mpar, ddim_steps=100, device=dev, rank=r, world=world, deterministic=False) for r in range(world)]
        z = _run_shards(shards, g['uc_s'], g['triples'], noise1, k, dev).cpu()
        mx = (z - g['z_steps'][k - 1]).abs().max().item()
        print('tuned shards (8 x 2 objects), %d DDIM step(s) vs the reference trajectory: max abs err %.2e' % (k, mx))
        assert torch.allclose(z, g['z_steps'][k - 1], atol=2e-2, rtol=2e-2), (k, mx)
        del shards
        torch.cuda.empty_cache()


def test_canonical_run_vs_reference_trajectory_O16(dev):
    """``deterministic=True`` at world 1: 16 objects on one GPU with the K cuts of the 4-object reference shard (the 256-row tiles
    take splits they would not choose for themselves) against the REFERENCE's own DDIMSampler (``shape_traj_full_O16``), and
    bit for bit against 4 canonical shards of 4 objects.  Bar: the product path's fp16-operand tolerance."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    g = load_golden('shape_traj_full_O16')
    df = DiffusionUNet(escfg.shape_unet_params(224))
    synth.seeded_fill_(df, prefix='unet3d_full.')
    mpar = escfg.shape_df_conf().model.params
    noise1 = synth.shape_noise(seed=7)
    z1 = ShapeDenoiser(df, mpar, ddim_steps=100, device=dev, deterministic=True).sample(g['uc_s'], g['triples'], noise1, n_steps=2)
    mx = (z1.cpu() - g['z_steps'][1]).abs().max().item()
    print('canonical arithmetic, 16 objects on one GPU, 2 DDIM steps vs the reference trajectory: max abs err %.2e' % mx)
    assert torch.allclose(z1.cpu(), g['z_steps'][1], atol=2e-2, rtol=2e-2), mx
    shards = [ShapeDenoiser(df, mpar, ddim_steps=100, device=dev, rank=r, world=4, deterministic=True) for r in range(4)]
    z4 = _run_shards(shards, g['uc_s'], g['triples'], noise1, 2, dev)
    assert torch.equal(z4, z1), 'max abs diff %.3e' % (z4 - z1).abs().max().item()


# ---- 'concat'-conditioned shape denoiser (sdfusion-txt2shape_concat_mp.yaml) ----
def _shape_concat(dev, mc, prefix, S):
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    df = DiffusionUNet(escfg.shape_unet_params(mc, concat=True), conditioning_key='concat')
    synth.seeded_fill_(df, prefix=prefix)
    return ShapeDenoiser(df, escfg.shape_df_conf(concat=True).model.params, ddim_steps=S, device=dev)


@pytest.mark.parametrize('tag,mc', [('tiny', 32), ('full', 224)])
def test_unet3d_concat_eps_vs_reference_golden(dev, tag, mc):
    """AttentionBlock / QKVAttentionLegacy, 5-channel input, stride-2 / nearest-x2 in all three axes."""
    g = load_golden('unet3d_concat_' + tag)
    den = _shape_concat(dev, mc, 'unet3d_concat_%s.' % tag, 100)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    eps = den.eps(g['x'], g['uc_s'], g['triples'], iteration=it, c=g['c_s'])
    e = _rel(eps, g['eps'])
    print('unet3d concat %s: fp16-MFMA eps vs fp32 reference golden: rel err %.3e' % (tag, e))
    assert e < 2e-2


def test_ddim_concat_tiny_loop_vs_reference_golden(dev):
    g = load_golden('unet3d_concat_tiny')
    den = _shape_concat(dev, 32, 'unet3d_concat_tiny.', 4)
    z = den.sample(g['uc_s'], g['triples'], synth.shape_noise(seed=7), c=g['c_s'])
    e = _rel(z, g['z_final'])
    print('ddim concat tiny: rel err %.3e' % e)
    assert e < 2e-2


def test_conv_down_dhw(dev):
    """Stride-2 conv in all three axes (ES_CONV_DOWN_DHW) against F.conv3d on fp16-rounded operands."""
    from echoscene_amd import hip
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    O, Cin, N, (D, H, W) = 2, 64, 48, (4, 4, 4)
    x = _rnd((O, Cin, 2 * D, 2 * H, 2 * W), 1).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3), 2) / np.sqrt(Cin * 27)).half().float()
    bias = _rnd((N,), 3)
    ref = F.conv3d(x, wt, bias, stride=2, padding=1)
    b = Builder(dev)
    xin = b.dev(_cl(x), torch.float16)
    out = b.buf(O * D * H * W, N, zero=True)
    b.conv(xin, PackedConv(wt, bias, dev), O, (D, H, W), mode=hip.CONV_DOWN_DHW, out_f32=out)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, _cl(ref)) < 1e-4


@pytest.mark.parametrize('O,dims,Cin,N,skipC', [(16, (16, 16, 16), 64, 224, 0), (32, (16, 8, 8), 96, 448, 32),
                                                 (33, (8, 4, 4), 64, 224, 0)])
def test_conv_ws_at_production_tile_counts(dev, O, dims, Cin, N, skipC):
    """The dominant kernel on launches shaped like the shipped ones (>= 256 tiles of 256 rows, W = 16 / 8 / 4, ragged last
    tile, bias + per-object vector + fp32 residual + both outputs, fused 1x1 skip phase) against F.conv3d on the same
    fp16-rounded operands (k_conv_ws: producer / consumer waves, the lane-owned-column epilogue with residual
    prefetch)."""
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    D, H, W = dims
    V = D * H * W
    if O * V < 256 * 256:
        from conftest import route_options
        if route_options().get('conv_force256') != '1':
            pytest.skip('needs the route option conv_force256 = 1 to reach the 256-row kernels at this size (run by test_conv_alternate_kernels)')
    x = _rnd((O, Cin) + dims, 1).half().float()
    wt = (_rnd((N, Cin, 3, 3, 3), 2) / np.sqrt(Cin * 27)).half().float()
    bias = _rnd((N,), 3)
    rowv = _rnd((O, N), 4)
    res = _rnd((O * V, N), 5)
    ref = F.conv3d(x, wt, bias, padding=1)
    b = Builder(dev)
    skip = None
    if skipC:
        xs = _rnd((O, skipC) + dims, 6).half().float()
        ws = (_rnd((N, skipC), 7) / np.sqrt(skipC)).half().float()
        ref = ref + F.conv3d(xs, ws[:, :, None, None, None])
        skip = (b.dev(_cl(xs), torch.float16), PackedConv(ws, None, dev))
    ref = _cl(ref) + rowv.repeat_interleave(V, 0) + res
    out = b.buf(O * V, N, zero=True)
    out16 = b.buf(O * V, N, dtype=torch.float16, zero=True)
    b.conv(b.dev(_cl(x), torch.float16), PackedConv(wt, bias, dev), O, dims, rowvec=View(b.dev(rowv)), res=b.dev(res),
           out_f32=out, out_f16=out16, skip=skip)
    b.finish().run()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 1e-4
    assert _rel(out16, ref) < 2e-3


def test_geglu_projection_at_4_objects_takes_the_producer_consumer_kernel(dev):
    """FeedForward GEGLU projection 448 -> 3584 at 16x8x8 with 4 objects: 16 row tiles x 16 column tiles = 256 workgroups, the one
    shape of the UNet that goes through k_conv_ws<GEGLU> (fewer objects: 128-row tiles; more: k_linear_ws).  Round 3 had a version
    of its epilogue that lost 16-byte pieces of the output now and then (buffer stores with an SGPR soffset the compiler reused right
    behind the store): the output is pre-filled with NaN and the launch repeated."""
    from echoscene_amd.plan import Builder
    from echoscene_amd.plan_vol import PackedConv
    from echoscene_amd import hip
    O, dims, K, N = 4, (16, 8, 8), 448, 3584
    M = O * dims[0] * dims[1] * dims[2]
    x = _rnd((M, K), 1).half()
    w = (_rnd((N, K), 2) / np.sqrt(K)).half().float()
    bias = 0.3 * _rnd((N,), 3)
    a_, g_ = (x.float() @ w.t() + bias).chunk(2, -1)
    ref = a_ * F.gelu(g_)
    b = Builder(dev)
    out = b.buf(M, N // 2, dtype=torch.float16)
    b.conv(b.dev(x, torch.float16), PackedConv(w, bias, dev, geglu=True), O, dims, out_f16=out, epilogue=hip.EPI_GEGLU, out_ld=N // 2)
    plan = b.finish()
    for rep in range(5):
        out.fill_(float('nan'))
        plan.run()
        torch.cuda.synchronize()
        assert _rel(out, ref) < 2e-3, rep


def test_conv_rowgroup_stats_feed_groupnorm(dev, monkeypatch):
    """es_conv_args.gn_stats_out: the conv leaves per (64-row group, column) sums of its fp32 output, and the GroupNorm that reads the
    tensor reduces them instead of passing over it.  Checks (a) the sums against the stored output for both ways they are formed
    (in the epilogue of the 256-row producer/consumer tiles -- 17 objects x 16^3 = 272 tiles, es_conv_emits_gn_stats() == 1; by
    k_rowgroup_stats behind a small problem, which the planner only asks for under ES_GN_RG_ANY=1), (b) GroupNorm(+SiLU) over
    [conv output | second conv output] with the sums against F.group_norm of the stored fp32 tensors, and (c) the same GroupNorm
    from a pass over the tensors (stats1 = NULL): equal within fp16 rounding.
    (`ES_CONV_WS=0` in test_conv_alternate_kernels runs this test with every sum coming from k_rowgroup_stats.)"""
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    from echoscene_amd import hip
    from echoscene_amd import plan_vol
    monkeypatch.setattr(plan_vol, 'VOL_GN_RG_ANY', True)
    for O, dims, Cin, N, epi in [(17, (16, 16, 16), 32, 224, 1), (2, (4, 8, 8), 64, 448, 0)]:
        D, H, W = dims
        V = D * H * W
        x = _rnd((O, Cin) + dims, 1)
        wa = (_rnd((N, Cin, 3, 3, 3), 2) / np.sqrt(Cin * 27)).half().float()
        wb = (_rnd((N, Cin, 1, 1, 1), 3) / np.sqrt(Cin)).half().float()
        ba = 0.3 * _rnd((N,), 4)
        rv = 0.5 * _rnd((O, N), 5)
        b = Builder(dev)
        xcl = b.dev(_cl(x), torch.float16)
        oa, ob = b.buf(O * V, N, zero=True), b.buf(O * V, N, zero=True)
        ia = b.conv(xcl, PackedConv(wa, ba, dev), O, dims, rowvec=View(b.dev(rv)), out_f32=oa)
        ib = b.conv(xcl, PackedConv(wb, None, dev), O, dims, res=oa, out_f32=ob)
        if not os.environ.get('ES_TEST_VOL_OPTIONS'):     # (the route options of test_conv_alternate_kernels re-route)
            import ctypes
            assert hip.lib().es_conv_emits_gn_stats(ctypes.byref(b.ops[ia].u.conv)) == epi
        ga, be = 1 + 0.1 * _rnd((2 * N,), 6), 0.1 * _rnd((2 * N,), 7)
        y = b.buf(O * V, 2 * N, dtype=torch.float16, zero=True)
        ig = b.groupnorm(oa, N, ob, N, O, V, b.dev(ga), b.dev(be), 1e-5, True, y)
        assert b.ops[ia].u.conv.gn_stats_out and b.ops[ib].u.conv.gn_stats_out and b.ops[ig].u.gn.stats1 and b.ops[ig].u.gn.stats2
        sa, sb = b._rg_stats[oa.data_ptr()][1], b._rg_stats[ob.data_ptr()][1]       # (producer op, sums)
        b.finish().run()
        torch.cuda.synchronize()
        for o_, s_ in ((oa, sa), (ob, sb)):
            g64 = o_.double().view(O * V // 64, 64, N)
            st = s_.view(2, O * V // 64, N).double()
            assert _rel(st[0], g64.sum(1)) < 1e-5 and _rel(st[1], (g64 * g64).sum(1)) < 1e-5
        cat = torch.cat([oa.view(O, V, N), ob.view(O, V, N)], 2).permute(0, 2, 1).reshape(O, 2 * N, D, H, W).cpu()
        ref = F.silu(F.group_norm(cat, 32, ga, be, 1e-5))
        assert _rel(y, _cl(ref)) < 2e-3
        hip.check(hip.lib().es_vol_set_option(b'gn_rg', 0), 'es_vol_set_option')     # (read by the planner at build time: this GroupNorm passes over the tensors)
        b2 = Builder(dev)
        y2 = b2.buf(O * V, 2 * N, dtype=torch.float16, zero=True)
        i2 = b2.groupnorm(oa, N, ob, N, O, V, b2.dev(ga), b2.dev(be), 1e-5, True, y2)
        assert not b2.ops[i2].u.gn.stats1
        b2.finish().run()
        torch.cuda.synchronize()
        hip.check(hip.lib().es_vol_set_option(b'gn_rg', 1), 'es_vol_set_option')
        assert (y.float() - y2.float()).abs().max() <= 2e-3 * max(1.0, float(y2.float().abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('O,dims,Cin,N,taps,res', [(4, (16, 16, 16), 224, 224, 27, True), (4, (16, 4, 4), 672, 672, 27, False),
                                                  (3, (16, 8, 8), 448, 448, 27, True), (2, (16, 4, 4), 672, 672, 27, True)])
def test_splitk_reduction_leaves_the_next_groupnorm_partials(dev, monkeypatch, O, dims, Cin, N, taps, res):
    """es_conv_args.gn_part_out: a conv launch split over K forms, in its reduction kernel, the per-tile partial sums of the
    GroupNorm that reads its output next; that GroupNorm then runs its apply kernel only.  The fused kernel repeats the statistics
    pass's summation order: conv output and GroupNorm output are BIT-identical to the plan with the two separate launches."""
    import ctypes
    from echoscene_amd.plan import Builder, View
    from echoscene_amd.plan_vol import PackedConv
    from echoscene_amd import hip, plan_vol
    if os.environ.get('ES_TEST_VOL_OPTIONS'):
        pytest.skip('route options re-route the launch')
    D, H, W = dims
    V = D * H * W
    k = 3 if taps == 27 else 1
    x = _rnd((O, Cin) + dims, 1)
    w = (_rnd((N, Cin, k, k, k), 2) / np.sqrt(Cin * taps)).half().float()
    bias, rv, r0 = 0.3 * _rnd((N,), 4), 0.5 * _rnd((O, N), 5), _rnd((O * V, N), 8)
    ga, be = 1 + 0.1 * _rnd((N,), 6), 0.1 * _rnd((N,), 7)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(plan_vol, 'VOL_GN_PART_FUSED', fused)
        b = Builder(dev)
        xcl = b.dev(_cl(x), torch.float16)
        o32 = b.buf(O * V, N, zero=True)
        ic = b.conv(xcl, PackedConv(w, bias, dev), O, dims, rowvec=View(b.dev(rv)), res=b.dev(r0) if res else None, out_f32=o32)
        y = b.buf(O * V, N, dtype=torch.float16, zero=True)
        ig = b.groupnorm(o32, N, None, 0, O, V, b.dev(ga), b.dev(be), 1e-5, True, y)
        cv, gn = b.ops[ic].u.conv, b.ops[ig].u.gn
        if fused:
            assert hip.lib().es_conv_emits_gn_part(ctypes.byref(cv)) == 1, 'this launch is expected to split K'
            assert cv.gn_part_out and gn.part_in == cv.gn_part_out and cv.gn_part_groups == 32
        else:
            assert not cv.gn_part_out and not gn.part_in
        b.finish().run()
        torch.cuda.synchronize()
        outs.append((o32.clone(), y.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    cl = outs[0][0].view(O, V, N).permute(0, 2, 1).reshape(O, N, D, H, W).cpu()
    assert _rel(outs[0][1], _cl(F.silu(F.group_norm(cl, 32, ga, be, 1e-5)))) < 2e-3


@pytest.mark.parametrize('env', [{'ES_TEST_VOL_OPTIONS': 'conv_ws=0'}, {'ES_TEST_VOL_OPTIONS': 'conv_tile=128'}, {'ES_TEST_VOL_OPTIONS': 'conv_force256=1'},
                                 {'ES_TEST_VOL_OPTIONS': 'conv_wssplit=0'}, {'ES_CONV_LINWS': '0'},
                                 # round 6: the few-objects kernels forced for EVERY eligible launch -- producer/consumer tiles of 64 and
                                 # 128 rows, and K split inside the workgroup (4 streams x 112 columns, 2 streams x 224 columns)
                                 {'ES_TEST_VOL_OPTIONS': 'conv_st_bm=64'}, {'ES_TEST_VOL_OPTIONS': 'conv_st_bm=128,conv_st_np=8'},
                                 {'ES_TEST_VOL_OPTIONS': 'conv_kw_ks=4'}, {'ES_TEST_VOL_OPTIONS': 'conv_kw_ks=2'},
                                 # k_conv_ws3 (A tile of a (chunk, kd, kh) group staged once, kw = -1 / +1 operands shifted in registers) for every
                                 # 3x3x3 SAME conv, small ones included (conv_force256), W = 4 / 8 / 16
                                 # (the default since its A/B; ES_CONV_A3=0 = k_conv_ws for those launches: both must give the goldens' bits)
                                 {'ES_TEST_VOL_OPTIONS': 'conv_force256=1', 'ES_CONV_A3': '1'}, {'ES_CONV_A3': '0'}])
def test_conv_alternate_kernels(env):
    """The conv dispatcher's other routes (the non-specialised k_conv_lean for 256-row tiles, 128-row tiles forced, small problems on
    128- / 64-row tiles with split-K instead of 256-row producer/consumer tiles with split-K) must give the same results: the conv unit tests and the full-width UNet golden test are re-run in a subprocess with the A/B switch set
    (route options are process-wide: tests/conftest.py applies ES_TEST_VOL_OPTIONS through es_vol_set_option; ES_CONV_LINWS is a
    timing-only environment switch).  conv_force256 routes EVERY conv of those tests (ragged, strided, up-sampled,
    1x1, fused skip, GEGLU) through the 256-row producer/consumer kernels, which otherwise only see launches with >= 256
    tiles."""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    here = os.path.dirname(os.path.abspath(__file__))
    sel = 'test_conv_mfma or test_conv_fused_skip or unet3d_full_eps or vqvae or test_conv_ws_at'
    sel += ' or test_conv_down_dhw or rowgroup_stats'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_hip_vol.py'), '-m', 'gpu', '-q', '-x', '-k', sel],
                       env=e, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('fam', ['crossattn', 'concat'])
def test_unet3d_without_message_passing_vs_reference_golden(dev, fam):
    """Shape denoisers of the configs without echo message passing (objects independent, c_s = key / concat channel)."""
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    g = load_golden('unet3d_nomp_' + fam)
    p = escfg.shape_unet_params(32, concat=(fam == 'concat'), mp=False)
    if fam == 'crossattn':
        p['context_dim'] = 64
    df = DiffusionUNet(p, conditioning_key=fam)
    synth.seeded_fill_(df, prefix='unet3d_nomp_%s.' % fam)
    den = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=100, device=dev)
    it = int(np.nonzero(den.sched.timesteps == int(g['t'][0]))[0][0])
    e = _rel(den.eps(g['x'], g['uc_s'], g['triples'], iteration=it, c=g['c_s']), g['eps'])
    den4 = ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev)
    ez = _rel(den4.sample(g['uc_s'], g['triples'], synth.shape_noise(seed=7), c=g['c_s']), g['z_final'])
    print('unet3d no-mp %s: eps rel err %.3e, 4-step DDIM rel err %.3e' % (fam, e, ez))
    assert e < 2e-2 and ez < 2e-2


def test_shards_without_message_passing(dev, monkeypatch):
    """No echo message passing: ranks never exchange anything during the loop; each shard's latents == the rows of the
    unsharded run (the final all-gather is replaced by a recorder)."""
    from echoscene_amd import parallel
    from echoscene_amd.model.unet import DiffusionUNet
    from echoscene_amd.samplers import ShapeDenoiser
    g = load_golden('unet3d_nomp_crossattn')
    p = escfg.shape_unet_params(32, mp=False)
    p['context_dim'] = 64
    df = DiffusionUNet(p, conditioning_key='crossattn')
    synth.seeded_fill_(df, prefix='unet3d_nomp_crossattn.')
    noise1 = synth.shape_noise(seed=7)
    mk = lambda r, w: ShapeDenoiser(df, escfg.shape_df_conf().model.params, ddim_steps=4, device=dev, rank=r, world=w)
    z_ref = mk(0, 1).sample(g['uc_s'], g['triples'], noise1, c=g['c_s'])
    got = []
    monkeypatch.setattr(parallel, 'all_gather_rows', lambda local, n, world, group=None: got.append(local.clone()) or local)
    for r in range(2):
        mk(r, 2).sample(g['uc_s'], g['triples'], noise1, c=g['c_s'])
    z = torch.cat(got, 0)
    assert z.shape == z_ref.shape and _rel(z, z_ref) < 2e-3
