"""Fixed cost vs per-K-step cost of the 256-row-tile conv kernel: same M, N, varying Cin (K steps = 27 * Cin / 32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
O, dims, cout = 32, (16, 16, 16), 224
M = O * 4096
pts = []
for cin in (32, 64, 96, 128, 224, 448):
    for taps in (27, 1):
        b = Builder(dev)
        x = b.buf(M, cin, dtype=torch.float16); x.normal_()
        w = torch.randn(cout, cin, 3, 3, 3) if taps == 27 else torch.randn(cout, cin)
        pc = PackedConv(w / (cin * taps) ** 0.5, torch.zeros(cout), dev)
        out = b.buf(M, cout); res = b.buf(M, cout); res.normal_()
        for _ in range(8):
            b.conv(x, pc, O, dims, res=res, out_f32=out)
        plan = b.finish()
        plan.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 8
        nks = taps * cin // 32
        print('Cin %4d taps %2d: K steps %4d  %7.1f us  (%.3f us per K step per round if fixed cost were 0)' % (cin, taps, nks, us, us / nks / 2), flush=True)
