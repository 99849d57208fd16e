"""Where does the per-tile fixed cost of k_conv_ws go?  3x3x3 conv at 16^3 x 32 objects (512 tiles of 256 rows = 2 rounds), N = 224, for
Cin in {32, 224, 448} (27 / 189 / 378 K units per tile) x epilogue in {fp32 out + fp32 residual, fp32 out, fp16 out}: a fit over Cin gives the
cost per K unit and the intercept per epilogue kind."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
O, dims, N = 32, (16, 16, 16), 224
M = O * 4096
res_t = {}
for kind in ('f32+res', 'f32', 'f16'):
    for cin in (32, 224, 448):
        b = Builder(dev)
        x = b.buf(M, cin, dtype=torch.float16); x.normal_()
        pc = PackedConv(torch.randn(N, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.zeros(N), dev)
        o32 = b.buf(M, N) if kind != 'f16' else None
        o16 = b.buf(M, N, dtype=torch.float16) if kind == 'f16' else None
        res = b.buf(M, N) if kind == 'f32+res' else None
        if res is not None:
            res.normal_()
        for _ in range(20):
            b.conv(x, pc, O, dims, res=res, out_f32=o32, out_f16=o16)
        plan = b.finish()
        plan.run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        res_t[(kind, cin)] = best
    t1, t2, t3 = res_t[(kind, 32)], res_t[(kind, 224)], res_t[(kind, 448)]
    per_unit = (t3 - t2) / (2 * (378 - 189))                 # us per K unit and tile (2 rounds of tiles per launch)
    icpt = (t2 - 2 * 189 * per_unit) / 2
    print('%-8s  Cin 32: %6.1f us   Cin 224: %6.1f us   Cin 448: %6.1f us   -> %.3f us per K unit, %.1f us per tile besides the K loop '
          '(Cin = 32 launch: %.1f us per round)' % (kind, t1, t2, t3, per_unit, icpt, t1 / 2), flush=True)
