"""Same-box A/B of the folded one-token self-attention (plan.ROWS_FOLD_ATTN1: the feed-forward launch forms t2 in its prologue,
ES_PRO_LN_ATTN, 11 dependent launches less per layout step) on configs[1] (32 nodes, model_channels 512): both modes in ONE process,
each with its own plan and captured graph; median of `reps` timings of `steps` replayed steps.  The two modes differ in arithmetic
(W1 (Wp x) against (W1 Wp) x): the maximum difference of a seeded 50-step run is printed instead of a bit comparison.
usage: python tools/ab_layout_fold.py [steps] [reps] [modes, e.g. 1,0,1,0]
A mode may carry other planner constants: "1:ROWS_LN_SPLIT=1" (fold on, the input projection in one slice)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from echoscene_amd import plan, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = sys.argv[3].split(',') if len(sys.argv) > 3 else ['1', '0', '1', '0']
defaults = {}
dev = torch.device('cuda')
O = 32
net, den, obj_embed, triples = bench.build_layout(dev, O, seed=100)          # (weights are built with the folded matrices present)
noise = synth.layout_noise(O, 8, 50)
ref = None
for mode_s in modes:
    parts = mode_s.split(':')
    mode = int(parts[0])
    for k, v in defaults.items():
        setattr(plan, k, v)
    for kv in parts[1:]:
        k, v = kv.split('=')
        defaults.setdefault(k, getattr(plan, k))
        setattr(plan, k, int(v))
    plan.ROWS_FOLD_ATTN1 = bool(mode)
    den._plans.clear()
    x = den.sample(obj_embed, triples, noise=noise, n_steps=50, use_graph=True).cpu()
    if ref is None:
        ref = x
    st = next(iter(den._plans.values()))
    ts = []
    for _ in range(reps):
        st['noise'].normal_()
        st['x'].copy_(st['noise'][0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st['plan'].sample(st['step'], 0, steps, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    print('fold=%-22s ops=%d launches=%s  step %.1f us (min %.1f max %.1f)  %.1f steps/s  max |x - x(first mode)| after 50 steps %.3e'
          % (mode_s, st['plan'].n_ops, getattr(st['plan'], 'n_launches', '?'), statistics.median(ts), min(ts), max(ts), 1e6 / statistics.median(ts),
             float((x - ref).abs().max())), flush=True)
