#!/usr/bin/env python
"""Scene calls over a sequence of scene sizes through the drop-in API (the reference's eval loop walks scenes of different
object counts): first call at a new size (plan build + graph capture) against the repeated call.
usage: python tools/scene_sizes_latency.py [sizes, default 32,10,16,10,6,16]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoscene_amd import synth, config as escfg
from model.SGDiff import SGDiff

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '32,10,16,10,6,16').split(',')]
opt = escfg.default_diff_opt('cuda', concat=False)
m = SGDiff('echoscene', opt, synth.VOCAB, replace_latent=False, with_changes=True, residual=True, gconv_pooling='avg',
           with_angles=True, clip=True, separated=False)
synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), prefix='lat.diff.')
synth.seeded_fill_(m.diff.ShapeDiff.df, prefix='lat.df.')
synth.seeded_fill_(m.diff.ShapeDiff.vqvae, prefix='lat.vq.')
m.diff.optimizer_ini()
m.cuda()
m.eval()
for k, O in enumerate(sizes):
    objs, triples = synth.synthetic_graph(O, seed=20 + k)
    tf, rf = synth.synthetic_features(O, triples.shape[0], seed=20 + k)
    args = (objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda())
    ts = []
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = m.sample_box_and_shape(*args, gen_shape=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('O = %2d (T = %3d): first call %.3f s, repeated %.3f s, finite %s' % (O, triples.shape[0], ts[0], ts[1],
          bool(torch.isfinite(d['shapes']).all())), flush=True)
