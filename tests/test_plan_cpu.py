"""Host-side checks of the layout planner (no GPU): the op list of one UNet1D step is emitted with a CPU Builder and examined from
its pointers alone (tools/plan_dryrun.py).  Since round 4 the head of the trunk rides on the launches of the GCN chain
(plan.Rider / Builder.ride): the problems of one launch must be independent, every operand must come from an earlier launch, and
riding must not change any product (same K slices: same bits)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _emit(monkeypatch, mode, **kw):
    import plan_dryrun
    from echoscene_amd import plan
    monkeypatch.setattr(plan, 'ROWS_RIDE', mode)
    b, _ = plan_dryrun.emit_layout_step_cpu(**kw)
    return plan_dryrun, b


def _products(b):
    from echoscene_amd import hip
    out = []
    for op in b.ops:
        if op.kind == hip.OP_LINEAR:
            a = op.u.linear
            out.append((a.M, a.K, a.N, a.kb_per_slice, a.act, a.nseg, a.wpack, a.out, a.out_slab_stride,
                        tuple((a.seg[s].ptr, a.seg[s].nslab, a.seg[s].pro, a.seg[s].mode) for s in range(a.nseg))))
    return out


@pytest.mark.parametrize('variant', [dict(), dict(enable_t_emb=False), dict(concat=True)], ids=['crossattn', 'no_t_emb', 'concat'])
def test_layout_step_launch_groups_are_independent_and_ordered(monkeypatch, variant):
    counts = {}
    for mode in (0, 1, 2):
        dry, b = _emit(monkeypatch, mode, mc=128, O=8, **variant)
        n, problems = dry.check(b)
        assert problems == [], problems
        counts[mode] = n
        for _, grp in dry.launches(b):
            assert len(grp) <= 3
    if variant.get('concat'):
        # the first trunk product already reads the GCN output: nothing can ride
        assert counts[0] == counts[1] == counts[2]
    else:
        # conv_in on the box embedding's launch + 5 (mode 1) / 8 (mode 2) trunk products on GCN launches
        assert counts[1] == counts[0] - 6 and counts[2] == counts[0] - 9, counts


def test_riding_changes_the_order_of_the_ops_only(monkeypatch):
    """same products (shape, K slices, epilogue, operand slab counts, prologues) whatever rides where: only the order differs"""
    from collections import Counter
    ref = None
    for mode in (0, 1, 2):
        _, b = _emit(monkeypatch, mode, mc=128, O=8)
        sig = Counter((p[:6] + tuple((s[1], s[2], s[3]) for s in p[9])) for p in _products(b))
        if ref is None:
            ref = sig
        assert sig == ref


def test_the_checker_sees_a_broken_order(monkeypatch):
    """negative control: a trunk product moved in front of the launch that produces its operand is reported"""
    from echoscene_amd import hip
    dry, b = _emit(monkeypatch, 0, mc=128, O=8)
    lin = [i for i, op in enumerate(b.ops) if op.kind == hip.OP_LINEAR]
    i, j = lin[-2], lin[-1]                     # the last ResBlock product and the output conv that reads it
    b.ops[i], b.ops[j] = b.ops[j], b.ops[i]
    _, problems = dry.check(b)
    assert problems
    # ... and two dependent products marked as ONE launch are reported as well
    b.ops[i], b.ops[j] = b.ops[j], b.ops[i]
    b.ops[i].u.linear.fuse_next = 1
    _, problems = dry.check(b)
    assert any('same launch' in p for p in problems), problems


def test_plan_launch_count_matches_the_runtime_grouping(monkeypatch):
    """Plan.n_launches (the bench line's launches_per_step) counts what es_plan_run launches: fused groups once"""
    from echoscene_amd import plan
    dry, b = _emit(monkeypatch, 2, mc=128, O=8)

    class _P:
        _arr = b.ops
    assert plan.Plan.n_launches.fget(_P) == len(dry.launches(b))


@pytest.mark.parametrize('consts', [dict(ROWS_SKIP_EARLY=True), dict(ROWS_GCN_SLICES=1), dict(ROWS_GCN_SLICES=4),
                                    dict(ROWS_LN_SPLIT=1, ROWS_VO1_SLICES=1, ROWS_CAV_SLICES=2)],
                         ids=['skip_early', 'gcn_1_slice', 'gcn_4_slices', 'other_slices'])
def test_other_planner_constants_keep_the_order_valid(monkeypatch, consts):
    """the planner's K-slice / fusion constants (plan.ROWS_*: build constants, A/B'd in round 3) combined with riding"""
    from echoscene_amd import plan
    for k, v in consts.items():
        monkeypatch.setattr(plan, k, v)
    dry, b = _emit(monkeypatch, 2, mc=128, O=8)
    n, problems = dry.check(b)
    assert problems == [], problems


def test_self_attention_fold_and_its_fallbacks(monkeypatch):
    """Round 5: the one-token self-attention of a transformer block is folded into its input projection where the library has the
    kernel (ES_PRO_LN_ATTN: model_channels <= 512) -- one launch less per block, the dependency check still green (the launch's
    ``res`` is an OUTPUT there: tools/plan_dryrun.py knows) -- and stays a launch of its own elsewhere (model_channels 1024, the
    planner constant off)."""
    from echoscene_amd import hip, plan
    counts = {}
    for mc, fold in ((128, True), (128, False), (1024, True)):
        monkeypatch.setattr(plan, 'ROWS_FOLD_ATTN1', fold)
        dry, b = _emit(monkeypatch, 2, mc=mc, O=8)
        n, problems = dry.check(b)
        assert problems == [], problems
        lin = [op.u.linear for op in b.ops if op.kind == hip.OP_LINEAR]
        n_attn = sum(1 for a in lin if a.seg[0].pro == hip.PRO_LN_ATTN)
        n_ln = sum(1 for a in lin if a.seg[0].pro == hip.PRO_LN)
        counts[(mc, fold)] = (n, n_attn, n_ln)
        for a in lin:
            if a.seg[0].pro == hip.PRO_LN_ATTN:
                assert a.nseg == 1 and a.res and a.res2 and not a.seg[0].gamma and not a.seg[0].beta and a.seg[0].gs == a.K
                assert hip.lib().es_linear_rows_takes_ln_attn(__import__('ctypes').byref(a)) == 1
    (n1, a1, l1), (n0, a0, l0), (nb, ab, lb) = counts[(128, True)], counts[(128, False)], counts[(1024, True)]
    assert a1 == 11 and l1 == 0 and a0 == 0 and l0 == 22 and n0 - n1 == 11, counts
    assert ab == 0 and lb == 22, counts                      # K = 1024: 8 k-blocks per wave, no formed-row kernel -> not folded


def test_groupnorm_widths_outside_the_register_prologue_are_separate_launches(monkeypatch):
    """model_channels = 384 (GroupNorm groups of 12 / 24 / 36 channels; reference: any channels % 32 == 0): the norms are OP_GN
    launches over whole matrices, no rows product carries a GroupNorm prologue or a K split, nothing rides or is folded."""
    from echoscene_amd import hip, plan
    assert plan.rows_gn_in_registers(512) and plan.rows_gn_in_registers(1024) and plan.rows_gn_in_registers(128)
    assert not plan.rows_gn_in_registers(384) and not plan.rows_gn_in_registers(768) and not plan.rows_gn_in_registers(1152)
    for concat in (False, True):
        dry, b = _emit(monkeypatch, 2, mc=384, O=8, concat=concat)
        n, problems = dry.check(b)
        assert problems == [], problems
        assert sum(1 for op in b.ops if op.kind == hip.OP_GN) >= 34
        for op in b.ops:
            if op.kind == hip.OP_LINEAR:
                a = op.u.linear
                assert a.kb_per_slice == 0                                   # no K-split slab outputs anywhere in this plan
                for s in range(a.nseg):
                    assert a.seg[s].pro not in (hip.PRO_GN, hip.PRO_GN_SILU, hip.PRO_LN_ATTN)


def _unpack(pl):
    """[N, K] matrix of a PackedLinear's MFMA-fragment image (es_pack_linear_f32: ((n tile, k block), lane, 4) with
    n = 16 nt + lane % 16, k = 16 kb + 4 (lane / 16) + e)"""
    import torch
    NT, KB = (pl.N + 15) // 16, (pl.K + 15) // 16
    img = pl.w.detach().cpu().reshape(NT, KB, 4, 16, 4)                  # [nt, kb, q, j, e]
    W = img.permute(0, 3, 1, 2, 4).reshape(NT * 16, KB * 16)             # [(nt, j), (kb, q, e)]
    return W[:pl.N, :pl.K].double()


def test_self_attention_fold_weights_reproduce_the_block_on_the_host():
    """The host half of the fold, without a GPU: with the folded weight images the planner builds -- [Wp ; W1 P Wp] for the input
    projection (GroupNorm affine inside), the cross-attention vector's matrix with the self-attention's bias in its bias --
    t2 = rstd(t0) u + t0 + cav, evaluated in float64 from the oracle's own block input and GCN output, is the oracle's
    attn1(norm1(t0)) + t0 + attn2(norm2(.), ctx) (attention.py:237-245 on one token) at every transformer block of the tiny denoiser."""
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from conftest import load_golden
    from echoscene_amd import synth, config as escfg, plan
    from echoscene_amd.model.unet import UNet1DModel
    from echoscene_amd.samplers import _cpu_sd
    from oracle import echoscene_oracle as orc
    assert plan.ROWS_FOLD_ATTN1
    g = load_golden('unet1d_tiny')
    kw = dict(escfg.layout_denoiser_kwargs(128))
    kw['concat_dim'] = kw['crossattn_dim'] = 128
    net = UNet1DModel(**kw)
    synth.seeded_fill_(net, prefix='unet1d_tiny.')
    sd = _cpu_sd(net)
    w = plan.UNet1DWeights(sd, net, torch.device('cpu'))
    trace = {}
    orc.unet1d_forward({k: v for k, v in sd.items()}, g['box'], g['obj_embed'], g['triples'], g['t'], trace=trace)
    ctx = trace['ctx'].double().reshape(g['box'].shape[0], -1)
    Wc, bc = _unpack(w.cav_all), w.cav_all.b.double()
    cav_all = ctx @ Wc.t() + bc
    n_blocks = 0
    for name, (k, C) in w.ca.items():
        d = w.items[name]
        assert 'proj_in_u' in d and d['proj_in_u'].N == 2 * C
        pre, idx = name.rsplit('.', 1)
        xin = trace['%s.%d' % (pre, int(idx) - 1)].double().reshape(-1, C)              # the ResBlock in front of the transformer
        xn = F.group_norm(xin.unsqueeze(-1), 32, None, None, 1e-6).squeeze(-1)          # (the affine sits in the weights)
        t0u = xn @ _unpack(d['proj_in_u']).t() + d['proj_in_u'].b.double()
        t0, u = t0u[:, :C], t0u[:, C:]
        rstd = 1.0 / torch.sqrt(t0.var(dim=1, unbiased=False, keepdim=True) + 1e-5)
        off = sum(c for n2, (k2, c) in w.ca.items() if k2 < k)
        t2 = rstd * u + t0 + cav_all[:, off:off + C]
        tb = name + '.transformer_blocks.0'
        ref0, ref2 = trace[tb + ':in'].double().reshape(-1, C), trace[tb + ':attn2'].double().reshape(-1, C)
        assert (t0 - ref0).abs().max() < 2e-5 * max(1.0, float(ref0.abs().max())), name
        assert (t2 - ref2).abs().max() < 5e-5 * max(1.0, float(ref2.abs().max())), (name, float((t2 - ref2).abs().max()))
        n_blocks += 1
    assert n_blocks == 11
