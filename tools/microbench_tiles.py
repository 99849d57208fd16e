"""Few objects per GPU: producer/consumer tiles of 64 / 128 rows (k_conv_ws<64|128, 4, NP>) against the dispatcher's own choice, per
layer shape of an O-object shard, with split K as listed; time per launch INCLUDING the split's reduction launch.  NW weight sets are
cycled so that weights stream as in a step.  usage: python tools/microbench_tiles.py --O 4"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv

ap = argparse.ArgumentParser()
ap.add_argument('--O', type=int, default=4)
ap.add_argument('--nw', type=int, default=6)
ap.add_argument('--shapes', default='all')
a = ap.parse_args()
dev = torch.device('cuda')
O = a.O
SH = [((16, 16, 16), 224, 224, 27), ((16, 16, 16), 448, 224, 27), ((16, 8, 8), 448, 448, 27), ((16, 8, 8), 896, 448, 27),
      ((16, 4, 4), 672, 672, 27), ((16, 4, 4), 1344, 672, 27),
      ((16, 8, 8), 448, 448, 1), ((16, 8, 8), 448, 1344, 1), ((16, 8, 8), 1792, 448, 1), ((16, 4, 4), 672, 672, 1), ((16, 4, 4), 672, 2016, 1),
      ((16, 4, 4), 2688, 672, 1)]
if a.shapes != 'all':
    SH = [SH[int(i)] for i in a.shapes.split(',')]
lib = hip.lib()


def setopt(k, v):
    hip.check(lib.es_vol_set_option(k.encode(), int(v)), 'es_vol_set_option')


for (dims, cin, cout, taps) in SH:
    D, H, W = dims
    M = O * D * H * W
    nks = taps * cin // 32
    ntn = (cout + 223) // 224
    pcs = []
    for i in range(a.nw):
        w = torch.randn(cout, cin, 3, 3, 3) if taps == 27 else torch.randn(cout, cin)
        pcs.append(PackedConv(w / (cin * taps) ** 0.5, torch.zeros(cout), dev))
    Z = {'conv_st_bm': 0, 'conv_st_np': 4, 'conv_st_ns': 3, 'conv_kw_ks': 0}
    cfgs = [('auto', {}, None, 0), ('exact', {}, None, 32), ('plainS1', {}, 1, 0), ('plainS2', {}, 2, 0), ('plainS4', {}, 4, 0)]
    for ks in (4, 2):                            # K split inside the workgroup (k_conv_kw)
        tiles = ((M + 63) // 64) * ntn * (2 if ks == 4 else 1)
        for S in (1, 2, 4):
            if S > 1 and (tiles * S > 640 or nks // (S * ks) < 4):
                continue
            cfgs.append(('kw%d/S%d' % (ks, S), {'conv_kw_ks': ks}, S, 0))
    for bm, np_, ns in ((64, 4, 3), (64, 4, 6), (128, 4, 3), (128, 8, 3), (128, 8, 5)):
        tiles = ((M + bm - 1) // bm) * ntn
        for S in (1, 2, 4, 8, 16):
            if S > 1 and (tiles * S > 640 or nks // S < 6):
                continue
            if S == 1 and tiles < 32:
                continue
            cfgs.append(('%d/%d/%d/S%d' % (bm, np_, ns, S), {'conv_st_bm': bm, 'conv_st_np': np_, 'conv_st_ns': ns}, S, 0))
    refs = {}
    gen = torch.Generator(device='cpu').manual_seed(7)
    x0 = torch.randn(M, cin, generator=gen).to(dev).half()
    res0 = torch.randn(M, cout, generator=gen).to(dev)
    line = '%-8s %4d->%4d t%2d M=%5d nks=%3d:' % ('x'.join(map(str, dims)), cin, cout, taps, M, nks)
    best = None
    for name, opts, S, oh in cfgs:
        for k, v in dict(Z, **opts).items():
            setopt(k, v)
        bm = -opts['conv_kw_ks'] if 'conv_kw_ks' in opts else opts.get('conv_st_bm', 0)
        b = Builder(dev)
        b.o_hint = oh
        x = b.buf(M, cin, dtype=torch.float16); x.copy_(x0)
        out = b.buf(M, cout)
        res = b.buf(M, cout); res.copy_(res0)
        for pc in pcs:
            b.conv(x, pc, O, dims, res=res, out_f32=out, splitk=S)
        plan = b.finish()
        plan.run(); torch.cuda.synchronize()
        us = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / a.nw)
        if name in ('plainS1', 'plainS2', 'plainS4'):
            refs[name] = out.clone()
        flag = ''
        if bm > 0 and S in (1, 2, 4):          # every conv kernel has the same K order and cuts: the same S must match bit for bit
            r = refs['plainS%d' % S]
            flag = '' if torch.equal(out, r) else '(!= plainS%d: max %.3g)' % (S, (out - r).abs().max().item())
        if bm < 0 and S == 1:                  # KS streams inside the workgroup == a split of S = KS over workgroups, bit for bit
            r = refs['plainS%d' % -bm]
            flag = '(== plainS%d)' % -bm if torch.equal(out, r) else '(!= plainS%d: max %.3g)' % (-bm, (out - r).abs().max().item())
        line += '  %s %.1f%s' % (name, us, flag)
        if bm != 0 and (best is None or us < best[1]):
            best = (name, us)
    for k, v in Z.items():
        setopt(k, v)
    fl = 2.0 * M * cout * cin * taps
    print(line + '   | best %s %.1f us = %.0f TF (%.1f GF)' % (best[0], best[1], fl / best[1] / 1e6, fl / 1e9), flush=True)
