import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from echoscene_amd import hip
from echoscene_amd.plan import Builder, PackedLinear, View, seg, norm_segs
dev = torch.device('cuda')
rs = np.random.RandomState(3)
T, O, H, N1 = int(sys.argv[1]), 32, 256, int(sys.argv[2])
t1 = torch.from_numpy(rs.standard_normal((3, T, H)).astype(np.float32))
Wb = torch.from_numpy((rs.standard_normal((N1, H)) / 16).astype(np.float32)); bb = torch.from_numpy(rs.standard_normal(N1).astype(np.float32))
x4 = torch.from_numpy(rs.standard_normal((4, O, 512)).astype(np.float32))
Wc = torch.from_numpy((rs.standard_normal((512, 512)) / 22).astype(np.float32)); bc = torch.from_numpy(rs.standard_normal(512).astype(np.float32))
ga = torch.from_numpy(1 + 0.1 * rs.standard_normal(512).astype(np.float32)); be = torch.from_numpy(0.1 * rs.standard_normal(512).astype(np.float32))
emb = torch.from_numpy(rs.standard_normal((1, 512)).astype(np.float32))
ref1 = F.relu(F.linear(F.relu(t1.sum(0)), Wb, bb))
xs = x4.sum(0)
ref2 = F.linear(F.silu(F.group_norm(xs.unsqueeze(-1), 32, ga, be, 1e-5).squeeze(-1)), Wc, bc) + emb
for fam in (1,):
    hip.lib().es_rows_set_kernel_family(fam)
    for fuse in (False, True):
        b = Builder(dev)
        t1d = b.dev(t1); x4d = b.dev(x4); gad, bed, embd = b.dev(ga), b.dev(be), b.dev(emb)
        tv = View(t1d[0], nslab=3, slab_stride=T * H)
        xv = View(x4d[0], nslab=4, slab_stride=O * 512)
        o1 = View(b.buf(T, N1))
        print('family', fam, 'fuse', fuse, 'building', flush=True)
        b.linear([seg(tv, pre_act=hip.ACT_RELU)], PackedLinear(Wb, bb, dev), T, o1, act=hip.ACT_RELU, fuse_next=fuse)
        mode = sys.argv[3]
        if mode == 'gn':
            o2 = b.linear(norm_segs([xv], gad, bed, 1e-5, True, C=512), PackedLinear(Wc, bc, dev), O, res=View(embd, ld=0, width=512))
        elif mode == 'gn_nores':
            o2 = b.linear(norm_segs([xv], gad, bed, 1e-5, True, C=512), PackedLinear(Wc, bc, dev), O)
        elif mode == 'plain':
            o2 = b.linear([seg(xv)], PackedLinear(Wc, bc, dev), O, res=View(embd, ld=0, width=512))
        else:
            o2 = b.linear([seg(xv)], PackedLinear(Wc, bc, dev), O, split=(False if mode == 'nosplit' else int(mode[1:]) if mode[0] == 's' else None))
        b.finish().run()
        torch.cuda.synchronize()
        print('   err n1b %.3e  rider %.3e (nslab %d)' % (float((o1.value().cpu() - ref1).abs().max()), float((o2.value().cpu() - ref2).abs().max()), o2.nslab), flush=True)
