"""Is the conv kernel clock/power-limited?  Same launch, random vs all-zero operands (identical instruction stream;
zero operands toggle far fewer bits in the MFMA / LDS / memory paths)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
O, dims, cin, cout = 32, (16, 16, 16), 224, 224
M = O * 4096
for tag, xs, ws in (('random', 1.0, 1.0), ('zero activations', 0.0, 1.0), ('zero weights', 1.0, 0.0), ('all zero', 0.0, 0.0), ('random again', 1.0, 1.0)):
    b = Builder(dev)
    x = b.buf(M, cin, dtype=torch.float16); x.normal_(); x.mul_(xs)
    pc = PackedConv(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5 * ws, torch.zeros(cout), dev)
    out = b.buf(M, cout); res = b.buf(M, cout); res.normal_()
    for _ in range(20):
        b.conv(x, pc, O, dims, res=res, out_f32=out)
    plan = b.finish()
    plan.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-18s %7.1f us  %7.1f TFLOP/s' % (tag, us, 2.0 * M * cout * cin * 27 / us / 1e6), flush=True)
