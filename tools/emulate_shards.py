#!/usr/bin/env python
"""Single-GPU emulation of rank 0 of an N-way object-sharded shape loop (no collective: the all-gather is
replaced by a local pad) -> predicted strong-scaling curve before the driver's 8-GPU run.
usage: python tools/emulate_shards.py [--nodes 32] [--steps 20]"""
import argparse, os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from echoscene_amd import parallel

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=32)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--worlds', default='1,2,4,8')
ap.add_argument('--deterministic', action='store_true', help='the canonical arithmetic (K splits of a 4-object reference shard at every world size, 1 included: bit-exact across world sizes)')
ap.add_argument('--tuned', action='store_true', help='every shard tunes K splits to its local object count (the default since round 6; kept for old command lines)')
ap.add_argument('--weak', action='store_true', help='batch of <world> scenes, one per rank (bench.py --scaling weak)')
a = ap.parse_args()
dev = torch.device('cuda', 0)
for _kv in [x for x in os.environ.get('ES_TOOL_VOL_OPTIONS', '').split(',') if x]:      # route options of this run: "name=value,..."
    from echoscene_amd import hip as _hip
    _hip.check(_hip.lib().es_vol_set_option(_kv.split('=')[0].encode(), int(_kv.split('=')[1])), 'es_vol_set_option')
O = a.nodes
net, lden, obj_embed, triples = bench.build_layout(dev, O, seed=100)


def fake_gather(local, num_rows, world, group=None, out=None):
    if out is None:
        out = torch.zeros((max(num_rows, local.shape[0]),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    out[:local.shape[0]].copy_(local)        # rank 0's block; the other ranks' rows stay as they are
    return out[:num_rows]


parallel.all_gather_rows = fake_gather
res = {}
for w in [int(x) for x in a.worlds.split(',')]:
    if a.weak:
        from echoscene_amd import synth
        _, triples = synth.collate_graphs([synth.synthetic_graph(a.nodes, seed=100 + s) for s in range(w)])
        O = a.nodes * w
    df, sden, uc = bench.build_shape(dev, O, 100, triples, 0, w, deterministic=a.deterministic)
    noise1 = torch.randn(1, 3, 16, 16, 16, device=dev)
    sden.sample(uc, triples, noise1=noise1, n_steps=3)
    ss = next(iter(sden._plans.values()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if w == 1:
        ss['plan'].sample(ss['step'], 0, a.steps)
    else:
        sden._cur, sden._use_graph = ss, True
        parallel.sharded_ddim_loop(sden, O, a.steps, w)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res[w] = dt * 1e3
    print('world %d: O_local %d  shape step %.3f ms  speed-up vs 1: %.2f' % (w, ss['hi'] - ss['lo'], dt * 1e3, res[1] / (dt * 1e3) if 1 in res else 0), flush=True)
    del df, sden, ss
    torch.cuda.empty_cache()
print(json.dumps(res))
