"""Per-op cost of ONE layout denoising step IN CONTEXT: the plan truncated after op k (k = 1 .. n), each prefix captured and replayed
as a hipGraph; t[k] - t[k-1] = what op k adds to the dependent chain with cold weights and its real operands.
usage: python tools/layout_op_times.py [O]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from echoscene_amd import hip
from echoscene_amd.plan import Builder

O = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)
den.sample(obj_embed, triples, noise=None, n_steps=3)
st = next(iter(den._plans.values()))
plan = st['plan']
ops = list(plan._arr)
PRO = {0: '-', 1: 'silu', 2: 'gn', 3: 'gn_silu', 4: 'ln', 5: 'geglu', 6: 'ln_attn'}


def sig(op):
    if op.kind != hip.OP_LINEAR:
        return 'op kind %d' % op.kind
    a = op.u.linear
    nkb = (a.K + 15) // 16
    S = (nkb + a.kb_per_slice - 1) // a.kb_per_slice if a.kb_per_slice else 1
    segs = ','.join('%d%s%s' % (a.seg[j].width, ('g', 'G', 'C', 'C')[a.seg[j].mode] if a.seg[j].mode else '', ('/' + PRO[a.seg[j].pro]) if a.seg[j].pro else '') +
                    ('x%d' % a.seg[j].nslab if a.seg[j].nslab > 1 else '') for j in range(a.nseg))
    return 'M%d K%d N%d S%d [%s]%s%s%s' % (a.M, a.K, a.N, S, segs, ' res%d' % max(a.res_nslab, 1) if a.res else '', ' act%d' % a.act if a.act else '',
                                           ' +fused' if a.fuse_next else '')


def time_prefix(k, reps=30):
    b = Builder(dev)
    b.ops, b.keep = ops[:k], plan.keep
    sub = b.finish()
    sub.sample(st['step'], 0, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sub.sample(st['step'], 0, reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


prev, agg = 0.0, collections.OrderedDict()
k = 1
while k <= len(ops):
    kk = k
    while kk <= len(ops) and ops[kk - 1].kind == hip.OP_LINEAR and ops[kk - 1].u.linear.fuse_next:
        kk += 1                                  # a fused group is one launch
    t = time_prefix(kk)
    name = ' || '.join(sig(ops[j]) for j in range(k - 1, kk))
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += t - prev
    prev = t
    k = kk + 1
print('whole step: %.1f us, %d launches' % (prev, sum(v[0] for v in agg.values())))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%7.1f us  n=%2d  avg %5.2f us  %s' % (t, n, t / n, name))
