"""Ablation micro-benchmark of k_conv_mfma (run with ES_CONV_DEBUG=<bits>: 1 no prefetch, 2 no MFMA, 4 no epilogue,
8 no LDS fragment reads).  Times single convs of the three UNet levels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv

dev = torch.device('cuda')
O = 32
for (dims, cin, cout, name) in [((16, 16, 16), 224, 224, 'L0 224->224'), ((16, 8, 8), 448, 448, 'L1 448->448'),
                                ((16, 4, 4), 672, 672, 'L2 672->672'), ((16, 4, 4), 1344, 672, 'L2 1344->672')]:
    D, H, W = dims
    M = O * D * H * W
    b = Builder(dev)
    x = b.buf(M, cin, dtype=torch.float16); x.normal_()
    pc = PackedConv(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.zeros(cout), dev)
    out = b.buf(M, cout)
    res = b.buf(M, cout); res.normal_()
    n = 10
    for _ in range(n):
        b.conv(x, pc, O, dims, res=res, out_f32=out)
    plan = b.finish()
    plan.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 2.0 * M * cout * cin * 27
    print('%-14s %8.1f us  %7.1f TFLOP/s' % (name, us, fl / us / 1e6), flush=True)
