"""Small-M regime of k_conv_mfma (few objects per GPU when a scene is sharded): time vs split-K for the layer
shapes of one O_local-object shard; NW distinct weight sets are cycled so that weights stream from HBM as in a
real step.  """
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv

ap = argparse.ArgumentParser()
ap.add_argument('--O', type=int, default=4)
ap.add_argument('--splits', default='-1,1,2,4,8,16')
ap.add_argument('--nw', type=int, default=12)
a = ap.parse_args()
dev = torch.device('cuda')
O = a.O
SH = [((16, 4, 4), 672, 672, 27), ((16, 4, 4), 1344, 672, 27), ((16, 4, 4), 672, 672, 1), ((16, 4, 4), 672, 5376, 1),
      ((16, 8, 8), 448, 448, 27), ((16, 8, 8), 448, 448, 1), ((16, 16, 16), 224, 224, 27), ((16, 16, 16), 448, 224, 27)]
for (dims, cin, cout, taps) in SH:
    D, H, W = dims
    M = O * D * H * W
    pcs = []
    for i in range(a.nw):
        w = torch.randn(cout, cin, 3, 3, 3) if taps == 27 else torch.randn(cout, cin)
        pcs.append(PackedConv(w / (cin * taps) ** 0.5, torch.zeros(cout), dev))
    line = '%-10s %4d->%4d t%2d M=%5d:' % ('x'.join(map(str, dims)), cin, cout, taps, M)
    for S in [int(x) for x in a.splits.split(',')]:
        b = Builder(dev)
        x = b.buf(M, cin, dtype=torch.float16); x.normal_()
        out = b.buf(M, cout)
        res = b.buf(M, cout); res.normal_()
        for pc in pcs:
            b.conv(x, pc, O, dims, res=res, out_f32=out, splitk=(None if S < 0 else S))
        plan = b.finish()
        plan.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.nw
        line += '  S=%2d %6.1f' % (S, us)
    fl = 2.0 * M * cout * cin * taps
    print(line + '   (us; %.1f GF)' % (fl / 1e9), flush=True)
