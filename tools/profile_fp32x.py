"""One shape step of the split-operand route (precision='fp32x') at O objects, for rocprofv3 --kernel-trace --stats.
usage: python tools/profile_fp32x.py [O] [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import synth, config as escfg
from echoscene_amd.samplers import ShapeDenoiser

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32x'
dev = torch.device('cuda')
_, triples = synth.synthetic_graph(O, seed=100)
df, sden, uc = bench.build_shape(dev, O, 100, triples)
den = ShapeDenoiser(df, escfg.shape_df_conf(224).model.params, ddim_steps=100, device=dev, precision=prec)
noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
den.sample(uc, triples, noise1=noise1, n_steps=1, use_graph=True)
ss = next(iter(den._plans.values()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ss['plan'].sample(ss['step'], 0, 3, use_graph=True); e1.record(); torch.cuda.synchronize()
print('%s O=%d: %.2f ms per shape step' % (prec, O, e0.elapsed_time(e1) / 3))
