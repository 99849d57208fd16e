"""Every op of rank 0's shape step when a 32-object scene is sharded over <world> GPUs (emulated on one GPU: no collective), each op
group timed on its own (20 launches back to back in one plan, HIP events): kind, shape, us per launch, launches per step, share.
usage: python tools/shard_op_table.py [--world 8] [--tuned]"""
import sys, os, collections, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import hip, synth, parallel
from echoscene_amd.plan import Builder

ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--nodes', type=int, default=32)
ap.add_argument('--tuned', action='store_true', help='(default)')
ap.add_argument('--deterministic', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda')
for _kv in [x for x in os.environ.get('ES_TOOL_VOL_OPTIONS', '').split(',') if x]:      # route options of this run: "name=value,..."
    from echoscene_amd import hip as _hip
    _hip.check(_hip.lib().es_vol_set_option(_kv.split('=')[0].encode(), int(_kv.split('=')[1])), 'es_vol_set_option')
O = a.nodes
_, triples = synth.synthetic_graph(O, seed=100)


def fake_gather(local, num_rows, world, group=None, out=None):
    if out is None:
        out = torch.zeros((max(num_rows, local.shape[0]),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    out[:local.shape[0]].copy_(local)
    return out[:num_rows]


parallel.all_gather_rows = fake_gather
df, sden, uc = bench.build_shape(dev, O, 100, triples, 0, a.world, deterministic=a.deterministic)
noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
sden.sample(uc, triples, noise1=noise1, n_steps=1, use_graph=True)
ss = next(iter(sden._plans.values()))
plan = ss['plan']
Ol = ss['hi'] - ss['lo']
groups = collections.OrderedDict()
for op in list(plan._arr):
    k = op.kind
    if k == hip.OP_CONV:
        c = op.u.conv
        key = ('conv', c.taps, c.Cin, c.Cin2 if c.a2 else 0, c.N, '%dx%dx%d' % (c.D, c.H, c.W), c.mode, c.epilogue, bool(c.res), bool(c.out_f32),
               bool(c.out_f16), bool(c.gn_stats_out), bool(c.gn_part_out))
        fl = 2.0 * c.O * c.D * c.H * c.W * c.N * (c.Cin * c.taps + (c.Cin2 if c.a2 else 0))
    elif k == hip.OP_GN:
        g = op.u.gn
        key = ('gn', g.C1, g.C2, g.V, g.silu, bool(g.raw_f16)); fl = 0
    elif k == hip.OP_LN:
        key = ('ln', op.u.ln.M, op.u.ln.C); fl = 0
    elif k == hip.OP_ATTN:
        t = op.u.attn
        key = ('attn', t.B, t.Ntok, t.heads, t.dhead); fl = 4.0 * t.B * t.heads * t.Ntok * t.Ntok * t.dhead
    elif k == hip.OP_LINEAR:
        l = op.u.linear
        key = ('rows', l.M, l.K, l.N); fl = 2.0 * l.M * l.K * l.N
    elif k in (hip.OP_FORK, hip.OP_JOIN):
        continue
    else:
        key = ('kind%d' % k,); fl = 0
    groups.setdefault(key, [fl, []])[1].append(op)
rows = []
for key, (fl, ops) in groups.items():
    b = Builder(dev)
    op0 = ops[0]
    lane0 = op0.lane
    op0.lane = 0
    b.ops, b.keep = [op0] * 20, plan.keep
    sub = b.finish()
    sub.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sub.run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    op0.lane = lane0
    rows.append((best * len(ops), len(ops), best, fl / best / 1e6 if fl else 0.0, key))
tot = sum(r[0] for r in rows)
print('# world %d (%d objects on this rank), %s shards: %d ops per step, summed stand-alone time %.3f ms'
      % (a.world, Ol, 'bit-exact' if a.deterministic else 'tuned', sum(r[1] for r in rows), tot / 1e3))
bykind = collections.OrderedDict()
for t, n, us, tf, key in rows:
    e = bykind.setdefault(key[0] + ('27' if key[0] == 'conv' and key[1] == 27 else ''), [0.0, 0])
    e[0] += t; e[1] += n
for kd, (t, n) in sorted(bykind.items(), key=lambda kv: -kv[1][0]):
    print('#   %-8s n=%3d  %8.1f us  %5.1f%%' % (kd, n, t, 100 * t / tot))
for t, n, us, tf, key in sorted(rows, key=lambda r: -r[0]):
    print('%5.1f%%  n=%2d  %7.1f us  %6.0f TF  %s' % (100 * t / tot, n, us, tf, key))
