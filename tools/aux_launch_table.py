"""GroupNorm / LayerNorm / attention launches of ONE shape step, each timed on its own (20 launches back to back in one plan,
HIP events): shape, us per launch, achieved GB/s on the ALGORITHMIC bytes (GroupNorm: 4 B read by the statistics pass + 4 B read and
2 B (+2 B raw copy) written by the apply pass; LayerNorm 4 + 2), share of the step.  usage: python tools/aux_launch_table.py [O]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import hip, synth
from echoscene_amd.plan import Builder

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
_, triples = synth.synthetic_graph(O, seed=100)
df, sden, uc = bench.build_shape(dev, O, 100, triples)
noise1 = torch.randn(1, 3, 16, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
sden.sample(uc, triples, noise1=noise1, n_steps=1, use_graph=True)
ss = next(iter(sden._plans.values()))
plan = ss['plan']
groups = collections.OrderedDict()
for op in list(plan._arr):
    if op.kind == hip.OP_GN:
        a = op.u.gn
        key = ('gn', a.C1, a.C2, a.V, a.silu, bool(a.raw_f16))
        nbytes = a.O * a.V * (a.C1 + a.C2) * (4 + 4 + 2 + (2 if a.raw_f16 else 0))
    elif op.kind == hip.OP_LN:
        a = op.u.ln
        key = ('ln', a.M, a.C)
        nbytes = a.M * a.C * 6
    elif op.kind == hip.OP_ATTN:
        a = op.u.attn
        key = ('attn', a.B, a.Ntok, a.heads, a.dhead)
        nbytes = 4 * a.B * a.heads * a.Ntok * a.Ntok * a.dhead        # FLOPs for attention
    else:
        continue
    groups.setdefault(key, [nbytes, []])[1].append(op)
rows = []
for key, (nbytes, ops) in groups.items():
    b = Builder(dev)
    b.ops, b.keep = [ops[0]] * 20, plan.keep
    sub = b.finish()
    sub.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sub.run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    rows.append((best * len(ops), len(ops), best, nbytes / best / 1e3, key))
tot = sum(r[0] for r in rows)
print('GroupNorm / LayerNorm / attention launches per step: %d, summed stand-alone time %.2f ms' % (sum(r[1] for r in rows), tot / 1e3))
for t, n, us, rate, key in sorted(rows, key=lambda r: -r[0]):
    unit = 'GFLOP/s' if key[0] == 'attn' else 'GB/s'
    print('%5.1f%%  n=%2d  %7.1f us  %8.0f %s  %s' % (100 * t / tot, n, us, rate, unit, key))
