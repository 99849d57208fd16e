#!/usr/bin/env python
"""End-to-end latency of one scene through the drop-in API at the shipped widths (config/full_mp.yaml equivalent,
seeded random weights): model.SGDiff.SGDiff('echoscene').sample_box_and_shape(gen_shape=True) = setup GCNs +
1000-step layout loop + 100-step DDIM shape loop + VQ-VAE decode to [O,1,64,64,64].
usage: python tools/e2e_latency.py [--nodes 32] [--concat]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoscene_amd import synth, config as escfg
from model.SGDiff import SGDiff

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=32)
ap.add_argument('--concat', action='store_true')
ap.add_argument('--prewarm', type=float, default=0.0, help='experiment: reserve this many GB in the caching allocator first (timed, reported)')
ap.add_argument('--host-weights', action='store_true', help='A/B: read the parameters through the host as before round 5')
ap.add_argument('--profile-first', action='store_true', help='cProfile of the first call (weight re-layouts, plan build, graph capture): top functions')
a = ap.parse_args()
opt = escfg.default_diff_opt('cuda', concat=a.concat)
m = SGDiff('echoscene', opt, synth.VOCAB, replace_latent=False, with_changes=True, residual=True, gconv_pooling='avg',
           with_angles=True, clip=True, separated=False)
synth.seeded_fill_(torch.nn.Module.state_dict(m.diff), prefix='lat.diff.')
synth.seeded_fill_(m.diff.ShapeDiff.df, prefix='lat.df.')
synth.seeded_fill_(m.diff.ShapeDiff.vqvae, prefix='lat.vq.')
m.diff.optimizer_ini()
m.cuda()
m.eval()
O = a.nodes
objs, triples = synth.synthetic_graph(O, seed=9)
tf, rf = synth.synthetic_features(O, triples.shape[0], seed=9)
args = (objs.cuda(), triples.cuda(), tf.cuda(), rf.cuda())
if a.host_weights:          # A/B: the round-4 route -- every parameter downloaded, folded on the host, uploaded again
    from echoscene_amd import samplers as _smp
    _smp.state_dict_for = lambda module, device=None: {k: v.detach().cpu() for k, v in module.state_dict().items()}
if a.prewarm > 0:
    t0 = time.perf_counter()
    blk = torch.empty(int(a.prewarm * (1 << 30)), dtype=torch.uint8, device='cuda')
    del blk
    torch.cuda.synchronize()
    print('prewarm %.1f GB: %.3f s' % (a.prewarm, time.perf_counter() - t0), flush=True)
for i in range(3):
    torch.cuda.synchronize()
    prof = None
    if a.profile_first and i == 0:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    d = m.sample_box_and_shape(*args, gen_shape=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats('tottime').print_stats(18)
        pstats.Stats(prof).sort_stats('cumulative').print_stats('echoscene_amd|model/', 45)
    print('call %d: %.3f s  (shapes %s, finite %s)' % (i, dt, tuple(d['shapes'].shape), bool(torch.isfinite(d['shapes']).all())), flush=True)

# ---- breakdown (each part synchronised) ----
diff = m.diff
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print('  %-28s %.3f s' % (name, time.perf_counter() - t0), flush=True)
    return r
oe, _, lat = timed('setup GCNs', lambda: diff._setup(*args, *args))
uc = timed('rel_s_mlp x2', lambda: (diff._rel_s(oe), diff._rel_s(lat)))
boxes = timed('layout loop (1000 steps)', lambda: diff._layout(args[1], oe, lat, None))
den = diff.ShapeDiff._denoiser()
z = timed('shape loop (100 DDIM steps)', lambda: den.sample(uc[0], args[1], noise1=torch.randn(1, 3, 16, 16, 16, device='cuda'),
                                                             c=uc[1] if a.concat else None))
sdf = timed('VQ-VAE decode', lambda: diff.ShapeDiff._decoder().decode_no_quant(z))

# ---- a NEW scene graph of the same size: plans are rebuilt and graphs re-captured ----
objs2, triples2 = synth.synthetic_graph(O, seed=10)
tf2, rf2 = synth.synthetic_features(O, triples2.shape[0], seed=10)
args2 = (objs2.cuda(), triples2.cuda(), tf2.cuda(), rf2.cuda())
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = m.sample_box_and_shape(*args2, gen_shape=True)
    torch.cuda.synchronize()
    print('new graph, call %d: %.3f s' % (i, time.perf_counter() - t0), flush=True)
