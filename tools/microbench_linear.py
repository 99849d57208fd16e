"""Transformer linears of the shape UNet on the conv kernels (1x1x1 'convs'): us and TFLOP/s per launch at O = 32.
Shapes: FeedForward GEGLU projection / output, qkv, at the 16x8x8 (C = 448) and 16x4x4 (C = 672) levels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
O = 32
for name, dims, K, N, geglu in (('ff1 GEGLU 448->3584 @16x8x8', (16, 8, 8), 448, 3584, True),
                                ('qkv 448->1344 @16x8x8', (16, 8, 8), 448, 1344, False),
                                ('ff2 1792->448 @16x8x8', (16, 8, 8), 1792, 448, False),
                                ('ff1 GEGLU 672->5376 @16x4x4', (16, 4, 4), 672, 5376, True),
                                ('qkv 672->2016 @16x4x4', (16, 4, 4), 672, 2016, False),
                                ('ff2 2688->672 @16x4x4', (16, 4, 4), 2688, 672, False)):
    M = O * dims[0] * dims[1] * dims[2]
    b = Builder(dev)
    x = b.buf(M, K, dtype=torch.float16); x.normal_()
    pc = PackedConv(torch.randn(N, K) / K ** 0.5, torch.zeros(N), dev, geglu=geglu)
    if geglu:
        out = b.buf(M, N // 2, dtype=torch.float16)
        for _ in range(20):
            b.conv(x, pc, O, dims, out_f16=out, epilogue=hip.EPI_GEGLU, out_ld=N // 2)
    else:
        out = b.buf(M, N, dtype=torch.float16)
        for _ in range(20):
            b.conv(x, pc, O, dims, out_f16=out)
    plan = b.finish()
    plan.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-30s %7.1f us  %7.1f TFLOP/s' % (name, us, 2.0 * M * K * N / us / 1e6), flush=True)
