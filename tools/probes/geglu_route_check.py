import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from echoscene_amd import hip
from echoscene_amd.plan import Builder
from echoscene_amd.plan_vol import PackedConv
dev = torch.device('cuda')
def rnd(shape, seed):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))
for O, dims, K, N in [(4, (16, 8, 8), 448, 3584), (2, (16, 8, 8), 448, 3584), (8, (16, 8, 8), 448, 3584), (32, (16, 8, 8), 448, 3584)]:
    M = O * dims[0] * dims[1] * dims[2]
    x = rnd((M, K), 1).half()
    w = (rnd((N, K), 2) / np.sqrt(K)).half().float()
    bias = 0.3 * rnd((N,), 3)
    h = x.float() @ w.t() + bias
    a_, g_ = h.chunk(2, -1)
    ref = a_ * F.gelu(g_)
    for rep in range(3):
        b = Builder(dev)
        out = b.buf(M, N // 2, dtype=torch.float16)
        out.fill_(float('nan'))
        b.conv(b.dev(x, torch.float16), PackedConv(w, bias, dev, geglu=True), O, dims, out_f16=out, epilogue=hip.EPI_GEGLU, out_ld=N // 2)
        b.finish().run(); torch.cuda.synchronize()
        o = out.float().cpu()
        nan = int(torch.isnan(o).sum())
        err = float((torch.nan_to_num(o) - ref).abs().max() / ref.abs().max())
        where = torch.nonzero(torch.isnan(o))[:6].tolist() if nan else []
        print('O %d rep %d: nan %d of %d, rel err %.2e %s' % (O, rep, nan, o.numel(), err, where), flush=True)
