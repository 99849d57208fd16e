"""Self-attention launches of the shape UNet on their own: us per call and TFLOP/s at O = 32 (1024 tokens x 8 heads x 56, 256 tokens x 8 x 84)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
dev = torch.device('cuda')
for name, B, N, H, dh in (('16x8x8: 1024 tok x 8 heads x 56', 32, 1024, 8, 56), ('16x4x4: 256 tok x 8 heads x 84', 32, 256, 8, 84)):
    b = Builder(dev)
    qkv = b.buf(B * N, 3 * H * dh, dtype=torch.float16); qkv.normal_()
    out = b.buf(B * N, H * dh, dtype=torch.float16)
    for _ in range(20):
        b.attention(qkv, B, N, H, dh, out)
    plan = b.finish()
    plan.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    print('%-36s %7.1f us  %6.0f TFLOP/s' % (name, best, 4.0 * B * H * N * N * dh / best / 1e6), flush=True)
