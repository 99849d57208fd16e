"""Every conv / linear launch of ONE shape step (O = 32, shipped widths), timed on its own (20 launches back to back in one
plan, HIP events): shape, dispatcher route, us, TFLOP/s, share of the conv time.  Shows which launch shapes hold
roofline.achieved down.  usage: python tools/conv_launch_table.py [O]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import hip, synth
from echoscene_amd.plan import Builder

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
for _kv in [x for x in os.environ.get('ES_TOOL_VOL_OPTIONS', '').split(',') if x]:      # route options of this run: "name=value,..."
    from echoscene_amd import hip as _hip
    _hip.check(_hip.lib().es_vol_set_option(_kv.split('=')[0].encode(), int(_kv.split('=')[1])), 'es_vol_set_option')
_, triples = synth.synthetic_graph(O, seed=100)
df, sden, uc = bench.build_shape(dev, O, 100, triples)
noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
sden.sample(uc, triples, noise1=noise1, n_steps=1, use_graph=True)
ss = next(iter(sden._plans.values()))
plan = ss['plan']
groups = collections.OrderedDict()
for op in list(plan._arr):
    if op.kind != hip.OP_CONV:
        continue
    c = op.u.conv
    key = (c.taps, c.Cin, c.Cin2 if c.a2 else 0, c.N, c.D, c.H, c.W, c.mode, c.epilogue, bool(c.res), bool(c.out_f32), bool(c.out_f16), bool(c.rowvec))
    groups.setdefault(key, []).append(op)
rows = []
for key, ops in groups.items():
    b = Builder(dev)
    b.ops, b.keep = [ops[0]] * 20, plan.keep
    sub = b.finish()
    sub.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sub.run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    taps, cin, cin2, N, D, H, W, mode, epi, res, o32, o16, rv = key
    M = O * D * H * W
    fl = 2.0 * M * N * (cin * taps + cin2)
    tiles = ((M + 255) // 256) * ((N + 223) // 224)
    rows.append((best * len(ops), len(ops), best, fl / best / 1e6, key, tiles))
tot = sum(r[0] for r in rows)
print('conv launches per step: %d, summed stand-alone time %.2f ms' % (sum(r[1] for r in rows), tot / 1e3))
for t, n, us, tf, key, tiles in sorted(rows, key=lambda r: -r[0]):
    taps, cin, cin2, N, D, H, W, mode, epi, res, o32, o16, rv = key
    print('%5.1f%%  n=%2d  %7.1f us  %6.0f TF  taps %2d  Cin %4d+%-4d N %4d  @%dx%dx%d mode %d epi %d res %d f32 %d f16 %d vec %d  tiles256 %d'
          % (100 * t / tot, n, us, tf, taps, cin, cin2, N, D, H, W, mode, epi, res, o32, o16, rv, tiles))
