"""One conv shape, one launch per plan run (for the ES_LEAN_ABL timing variants which printf per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
O, dims, cin, cout = 32, (16, 16, 16), 224, 224
if len(sys.argv) > 1 and sys.argv[1] == 'L1':
    dims, cin, cout = (16, 8, 8), 448, 448
D, H, W = dims
M = O * D * H * W
b = Builder(dev)
x = b.buf(M, cin, dtype=torch.float16); x.normal_()
pc = PackedConv(torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.zeros(cout), dev)
out = b.buf(M, cout); res = b.buf(M, cout); res.normal_()
b.conv(x, pc, O, dims, res=res, out_f32=out)
plan = b.finish()
for _ in range(3):
    plan.run(); torch.cuda.synchronize()
